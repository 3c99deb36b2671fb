!  lbfgsb_module -- the reference's public Fortran interface over the MI355X library.
!
!  Same module name, same public symbols (`setulb`, `lbfgsp_wp`) and the same
!  argument list as jacobwilliams/lbfgsb src/lbfgsb.f90:39-58, 88-89, so that the
!  reference's test/driver1.f90, driver2.f90 and driver3.f90 compile and link
!  unchanged.  Nothing is computed here: the call is forwarded through
!  iso_c_binding to `lbfgsb_hip_setulb_host_ik` (include/lbfgsb_hip.h), which runs
!  the iteration on the GPU.  Only the Fortran-specific marshalling lives here:
!  character(len=60) <-> char[60], logical <-> integer, optional file name.
!
!  Integer width: the module compiles unchanged with and without -fdefault-integer-8
!  (the build BASELINE.md section 3 calls mandatory at n = 1e8, src/lbfgsb.f90:246-265):
!  the width of the default INTEGER kind is handed to the library with every call.
!
!  -DREAL32 selects single precision exactly like the reference's
!  lbfgsb_kinds_module.F90:29-37 (REAL128 is not supported on the GPU).
      module lbfgsb_module

      use iso_c_binding
      use iso_fortran_env, only: output_unit, real32, real64

      implicit none

      private

#ifdef REAL32
      integer,parameter :: wp = real32
#else
      integer,parameter :: wp = real64
#endif
      integer,parameter,public :: lbfgsp_wp = wp

      public :: setulb
      public :: lbfgsb_release   ! extension: frees the GPU context of a run that was stopped
                                 ! by the caller (task = 'STOP...' without another setulb call,
                                 ! as test/driver2.f90:174-195 does); harmless otherwise

      ! default INTEGER / LOGICAL width of THIS compilation (4, or 8 under -fdefault-integer-8): the
      ! library takes nbd, iwa, lsave, isave as arrays of that width (lbfgsb_hip_setulb_host_ik), so
      ! the same source serves both builds
      integer,parameter :: ibytes = storage_size(1)/8

      interface
         function lbfgsb_hip_setulb_host_ik(n,m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,iprint,  &
                                            csave,lsave,isave,dsave,iteration_file,real_bytes, &
                                            mirror,int_bytes)                                  &
                                            bind(C,name='lbfgsb_hip_setulb_host_ik') result(rc)
            import :: c_int32_t, c_int64_t, c_double, c_char, c_ptr, wp
            integer(c_int64_t),value :: n, m, iprint
            integer(c_int32_t),value :: real_bytes, mirror, int_bytes
            real(wp) :: x(*), l(*), u(*), g(*), wa(*), dsave(*)
            real(wp) :: f
            real(c_double),value :: factr, pgtol
            integer :: nbd(*), iwa(*), lsave(*), isave(*)   ! default kind: int_bytes wide
            character(kind=c_char) :: task(*), csave(*)
            type(c_ptr),value :: iteration_file
            integer(c_int32_t) :: rc
         end function lbfgsb_hip_setulb_host_ik
         function lbfgsb_hip_release_host_ik(isave,int_bytes)                                  &
                                             bind(C,name='lbfgsb_hip_release_host_ik') result(rc)
            import :: c_int32_t
            integer :: isave(*)
            integer(c_int32_t),value :: int_bytes
            integer(c_int32_t) :: rc
         end function lbfgsb_hip_release_host_ik
         function lbfgsb_hip_last_error() bind(C,name='lbfgsb_hip_last_error') result(p)
            import :: c_ptr
            type(c_ptr) :: p
         end function lbfgsb_hip_last_error
      end interface

      contains

      subroutine setulb(n,m,x,l,u,Nbd,f,g,Factr,Pgtol,Wa,Iwa,Task, &
                        Iprint,Csave,Lsave,Isave,Dsave,iteration_file)

      integer,intent(in) :: n
      integer,intent(in) :: m
      real(wp),intent(inout) :: x(n)
      real(wp),intent(in) :: l(n)
      real(wp),intent(in) :: u(n)
      integer,intent(in) :: Nbd(n)
      real(wp),intent(inout) :: f
      real(wp),intent(inout) :: g(n)
      real(wp),intent(in) :: Factr
      real(wp),intent(in) :: Pgtol
      real(wp) :: Wa(*)
      integer :: Iwa(*)
      character(len=60),intent(inout) :: Task
      integer,intent(in) :: Iprint
      character(len=60) :: Csave
      logical :: Lsave(4)
      integer :: Isave(44)
      real(wp) :: Dsave(29)
      character(len=*),intent(in),optional :: iteration_file

      character(kind=c_char) :: ctask(60), ccsave(60)
      character(kind=c_char),allocatable,target :: cfile(:)
      integer :: clsave(4)
      integer(c_int32_t) :: rc
      type(c_ptr) :: pfile
      integer :: i, k

      do i = 1, 60
         ctask(i) = Task(i:i)
         ccsave(i) = Csave(i:i)
      end do
      do i = 1, 4
         clsave(i) = merge(1, 0, Lsave(i))
      end do
      pfile = c_null_ptr
      if (present(iteration_file)) then
         k = len_trim(iteration_file)
         allocate (cfile(k + 1))
         do i = 1, k
            cfile(i) = iteration_file(i:i)
         end do
         cfile(k + 1) = c_null_char
         pfile = c_loc(cfile)
      end if

      ! the library prints through C stdio: keep the two output streams in order
      if (Iprint >= 0) flush (output_unit)

      rc = lbfgsb_hip_setulb_host_ik(int(n, c_int64_t), int(m, c_int64_t), x, l, u, Nbd, f, g,        &
                                     real(Factr, c_double), real(Pgtol, c_double), Wa, Iwa, ctask,    &
                                     int(Iprint, c_int64_t), ccsave, clsave, Isave, Dsave, pfile,     &
                                     int(storage_size(1.0_wp)/8, c_int32_t), 0_c_int32_t,             &
                                     int(ibytes, c_int32_t))

      do i = 1, 60
         Task(i:i) = ctask(i)
         Csave(i:i) = ccsave(i)
      end do
      do i = 1, 4
         Lsave(i) = clsave(i) /= 0
      end do
      if (rc /= 0) then
         ! no GPU, no library, bad call sequence: there is no CPU path to fall back to
         write (output_unit,'(a,i0)') ' lbfgsb_hip_setulb_host failed, code ', rc
         Task = 'ERROR: LBFGSB_HIP FAILURE'
      end if

      end subroutine setulb

      subroutine lbfgsb_release(Isave)
      integer :: Isave(44)
      integer(c_int32_t) :: rc
      rc = lbfgsb_hip_release_host_ik(Isave, int(ibytes, c_int32_t))
      end subroutine lbfgsb_release

      end module lbfgsb_module
