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
      ! Extensions beside the unchanged setulb: the DEVICE-POINTER forms of include/lbfgsb_hip.h for a Fortran
      ! caller whose x, l, u, nbd, g live in HBM (type(c_ptr) from hipMalloc through iso_c_binding) and whose
      ! objective runs on the GPU -- nothing n-long crosses PCIe.  Same reverse-communication protocol, same task
      ! strings, same isave / dsave / lsave slots as setulb (src/lbfgsb.f90:88-244); the work arrays wa / iwa are
      ! replaced by a context handle (examples/driver_dev.f90 is driver2's loop, test/driver2.f90:66-195, on them).
      public :: lbfgsb_create, lbfgsb_destroy          ! context of n rows, m pairs (lbfgsb_hip_create / _destroy)
      public :: setulb_dev                             ! lbfgsb_hip_setulb_dev: x, g updated in place on the device
      public :: setulb_dev_pp                          ! lbfgsb_hip_setulb_dev_pp: ping-pong iterate buffers
      public :: lbfgsb_objective                       ! built-in device objectives (lbfgsb_hip_objective)
      public :: lbfgsb_set_option                      ! per-context switches (lbfgsb_hip_set_option): 'compact_w', ...
      public :: lbfgsb_error_message                   ! text of the last failure (lbfgsb_hip_last_error)
      ! flags of lbfgsb_create (include/lbfgsb_hip.h)
      integer,parameter,public :: LBFGSB_F_REAL32 = 1, LBFGSB_F_MIRROR_INDEX = 2, LBFGSB_F_NO_RETURN_SYNC = 4, &
                                  LBFGSB_F_PARALLEL_GCP = 8, LBFGSB_F_INDEX_TIES = 32, LBFGSB_F_DEFER_LNSRCH = 64
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
         ! ---- device-pointer forms (include/lbfgsb_hip.h) ----
         function lbfgsb_hip_create(n_local,n_global,row0,m,flags,device,stream,ctx)             &
                                    bind(C,name='lbfgsb_hip_create') result(rc)
            import :: c_int, c_int64_t, c_ptr
            integer(c_int64_t),value :: n_local, n_global, row0
            integer(c_int),value :: m, flags, device
            type(c_ptr),value :: stream
            type(c_ptr) :: ctx
            integer(c_int) :: rc
         end function lbfgsb_hip_create
         subroutine lbfgsb_hip_destroy(ctx) bind(C,name='lbfgsb_hip_destroy')
            import :: c_ptr
            type(c_ptr),value :: ctx
         end subroutine lbfgsb_hip_destroy
         function lbfgsb_hip_setulb_dev(ctx,x,l,u,nbd,f,g,factr,pgtol,task,iprint,csave,lsave,isave,dsave) &
                                        bind(C,name='lbfgsb_hip_setulb_dev') result(rc)
            import :: c_int, c_int32_t, c_double, c_char, c_ptr
            type(c_ptr),value :: ctx, x, l, u, nbd, g
            real(c_double) :: f, dsave(29)
            real(c_double),value :: factr, pgtol
            character(kind=c_char) :: task(*), csave(*)
            integer(c_int),value :: iprint
            integer(c_int32_t) :: lsave(4), isave(44)
            integer(c_int) :: rc
         end function lbfgsb_hip_setulb_dev
         function lbfgsb_hip_setulb_dev_pp(ctx,x0,x1,l,u,nbd,f,g0,g1,factr,pgtol,task,iprint,csave,lsave, &
                                           isave,dsave,cur) bind(C,name='lbfgsb_hip_setulb_dev_pp') result(rc)
            import :: c_int, c_int32_t, c_double, c_char, c_ptr
            type(c_ptr),value :: ctx, x0, x1, l, u, nbd, g0, g1
            real(c_double) :: f, dsave(29)
            real(c_double),value :: factr, pgtol
            character(kind=c_char) :: task(*), csave(*)
            integer(c_int),value :: iprint
            integer(c_int32_t) :: lsave(4), isave(44), cur
            integer(c_int) :: rc
         end function lbfgsb_hip_setulb_dev_pp
         function lbfgsb_hip_objective(ctx,kind,x,g,h_f) bind(C,name='lbfgsb_hip_objective') result(rc)
            import :: c_int, c_ptr
            type(c_ptr),value :: ctx, x, g, h_f
            integer(c_int),value :: kind
            integer(c_int) :: rc
         end function lbfgsb_hip_objective
         function lbfgsb_hip_set_option(ctx,name,val) bind(C,name='lbfgsb_hip_set_option') result(rc)
            import :: c_int, c_ptr, c_char, c_double
            type(c_ptr),value :: ctx
            character(kind=c_char) :: name(*)
            real(c_double),value :: val
            integer(c_int) :: rc
         end function lbfgsb_hip_set_option
         function c_strlen(s) bind(C,name='strlen') result(k)
            import :: c_ptr, c_size_t
            type(c_ptr),value :: s
            integer(c_size_t) :: k
         end function c_strlen
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

!  ---------------------------------------------------------------------------------------------------------
!  Device-pointer forms.  x, l, u, nbd, g are type(c_ptr) DEVICE addresses (real(wp) values, nbd 32-bit
!  integers, 16-byte aligned); f, task, csave, lsave, isave, dsave are ordinary host variables with setulb's
!  meaning slot for slot.  rc = 0, or the library's error code (text: lbfgsb_error_message()).
!  ---------------------------------------------------------------------------------------------------------

      subroutine lbfgsb_create(ctx, n, m, flags, rc, device)
      type(c_ptr),intent(out) :: ctx
      integer,intent(in) :: n, m
      integer,intent(in) :: flags            ! sum of LBFGSB_F_*; LBFGSB_F_REAL32 is added for a -DREAL32 build
      integer,intent(out) :: rc
      integer,intent(in),optional :: device
      integer(c_int) :: fl, dev
      fl = int(flags, c_int)
      if (storage_size(1.0_wp) == 32) fl = ior(fl, int(LBFGSB_F_REAL32, c_int))
      dev = 0
      if (present(device)) dev = int(device, c_int)
      ctx = c_null_ptr
      rc = lbfgsb_hip_create(int(n, c_int64_t), int(n, c_int64_t), 0_c_int64_t, int(m, c_int), fl, dev, &
                             c_null_ptr, ctx)
      end subroutine lbfgsb_create

      subroutine lbfgsb_destroy(ctx)
      type(c_ptr),intent(inout) :: ctx
      if (c_associated(ctx)) call lbfgsb_hip_destroy(ctx)
      ctx = c_null_ptr
      end subroutine lbfgsb_destroy

      subroutine marshal_in(Task, Csave, Lsave, Isave, Dsave, f, ctask, ccsave, l32, i32, d64, f64)
      character(len=60),intent(in) :: Task, Csave
      logical,intent(in) :: Lsave(4)
      integer,intent(in) :: Isave(44)
      real(wp),intent(in) :: Dsave(29), f
      character(kind=c_char),intent(out) :: ctask(60), ccsave(60)
      integer(c_int32_t),intent(out) :: l32(4), i32(44)
      real(c_double),intent(out) :: d64(29), f64
      integer :: i
      do i = 1, 60
         ctask(i) = Task(i:i)
         ccsave(i) = Csave(i:i)
      end do
      do i = 1, 4
         l32(i) = merge(1_c_int32_t, 0_c_int32_t, Lsave(i))
      end do
      i32 = int(Isave, c_int32_t)
      d64 = real(Dsave, c_double)
      f64 = real(f, c_double)
      end subroutine marshal_in

      subroutine marshal_out(Task, Csave, Lsave, Isave, Dsave, f, ctask, ccsave, l32, i32, d64, f64)
      character(len=60),intent(out) :: Task, Csave
      logical,intent(out) :: Lsave(4)
      integer,intent(out) :: Isave(44)
      real(wp),intent(out) :: Dsave(29), f
      character(kind=c_char),intent(in) :: ctask(60), ccsave(60)
      integer(c_int32_t),intent(in) :: l32(4), i32(44)
      real(c_double),intent(in) :: d64(29), f64
      integer :: i
      do i = 1, 60
         Task(i:i) = ctask(i)
         Csave(i:i) = ccsave(i)
      end do
      do i = 1, 4
         Lsave(i) = l32(i) /= 0
      end do
      Isave = int(i32, kind(Isave))
      Dsave = real(d64, wp)
      f = real(f64, wp)
      end subroutine marshal_out

      subroutine setulb_dev(ctx, x, l, u, Nbd, f, g, Factr, Pgtol, Task, Iprint, Csave, Lsave, Isave, Dsave, rc)
      type(c_ptr),intent(in) :: ctx
      type(c_ptr),intent(in) :: x, l, u, Nbd, g        ! device pointers
      real(wp),intent(inout) :: f
      real(wp),intent(in) :: Factr, Pgtol
      character(len=60),intent(inout) :: Task
      integer,intent(in) :: Iprint
      character(len=60) :: Csave
      logical :: Lsave(4)
      integer :: Isave(44)
      real(wp) :: Dsave(29)
      integer,intent(out) :: rc
      character(kind=c_char) :: ctask(60), ccsave(60)
      integer(c_int32_t) :: l32(4), i32(44)
      real(c_double) :: d64(29), f64
      call marshal_in(Task, Csave, Lsave, Isave, Dsave, f, ctask, ccsave, l32, i32, d64, f64)
      if (Iprint >= 0) flush (output_unit)
      rc = lbfgsb_hip_setulb_dev(ctx, x, l, u, Nbd, f64, g, real(Factr, c_double), real(Pgtol, c_double), ctask, &
                                 int(Iprint, c_int), ccsave, l32, i32, d64)
      call marshal_out(Task, Csave, Lsave, Isave, Dsave, f, ctask, ccsave, l32, i32, d64, f64)
      if (rc /= 0) Task = 'ERROR: LBFGSB_HIP FAILURE'
      end subroutine setulb_dev

      ! Ping-pong form: TWO buffers for x and two for g; cur (0 or 1) says which pair this return refers to --
      ! evaluate f at x(cur) into g(cur) on 'FG...', x(cur) / g(cur) are the iterate on 'NEW_X' and at the end
      ! (include/lbfgsb_hip.h: nothing is copied for t = x, r = g of src/lbfgsb.f90:2235-2236).
      subroutine setulb_dev_pp(ctx, x0, x1, l, u, Nbd, f, g0, g1, Factr, Pgtol, Task, Iprint, Csave, Lsave, Isave, &
                               Dsave, cur, rc)
      type(c_ptr),intent(in) :: ctx
      type(c_ptr),intent(in) :: x0, x1, l, u, Nbd, g0, g1     ! device pointers
      real(wp),intent(inout) :: f
      real(wp),intent(in) :: Factr, Pgtol
      character(len=60),intent(inout) :: Task
      integer,intent(in) :: Iprint
      character(len=60) :: Csave
      logical :: Lsave(4)
      integer :: Isave(44)
      real(wp) :: Dsave(29)
      integer,intent(out) :: cur, rc
      character(kind=c_char) :: ctask(60), ccsave(60)
      integer(c_int32_t) :: l32(4), i32(44), c32
      real(c_double) :: d64(29), f64
      call marshal_in(Task, Csave, Lsave, Isave, Dsave, f, ctask, ccsave, l32, i32, d64, f64)
      if (Iprint >= 0) flush (output_unit)
      c32 = 0
      rc = lbfgsb_hip_setulb_dev_pp(ctx, x0, x1, l, u, Nbd, f64, g0, g1, real(Factr, c_double),               &
                                    real(Pgtol, c_double), ctask, int(Iprint, c_int), ccsave, l32, i32, d64, c32)
      call marshal_out(Task, Csave, Lsave, Isave, Dsave, f, ctask, ccsave, l32, i32, d64, f64)
      cur = int(c32)
      if (rc /= 0) Task = 'ERROR: LBFGSB_HIP FAILURE'
      end subroutine setulb_dev_pp

      ! A switch of this context (include/lbfgsb_hip.h, lbfgsb_hip_set_option), before 'START': e.g.
      ! call lbfgsb_set_option(ctx, 'compact_w', 1.0d0, rc) -- the two passes over W read the W entries of the free
      ! variables only (the reference's Index(1:nfree) loops, src/lbfgsb.f90:1565-1583, 2743-2778; DESIGN.md 4g)
      subroutine lbfgsb_set_option(ctx, name, val, rc)
      type(c_ptr),intent(in) :: ctx
      character(len=*),intent(in) :: name
      real(c_double),intent(in) :: val
      integer,intent(out) :: rc
      character(kind=c_char) :: cname(len_trim(name) + 1)
      integer :: k
      do k = 1, len_trim(name)
         cname(k) = name(k:k)
      end do
      cname(len_trim(name) + 1) = c_null_char
      rc = lbfgsb_hip_set_option(ctx, cname, val)
      end subroutine lbfgsb_set_option

      ! Built-in objective on the context's stream: kind 0 = separable bounded quadratic (BASELINE.md 3),
      ! 1 = extended Rosenbrock (test/driver1.f90:274-289).  With f present the call waits for the value;
      ! without it the value stays on the device and the NEXT setulb_dev / setulb_dev_pp call (the 'FG' re-entry)
      ! brings it over with its own sums and stores it in its f argument: one host sync per evaluation less.
      subroutine lbfgsb_objective(ctx, kind, x, g, rc, f)
      type(c_ptr),intent(in) :: ctx, x, g
      integer,intent(in) :: kind
      integer,intent(out) :: rc
      real(wp),intent(out),optional :: f
      real(c_double),target :: f64
      if (present(f)) then
         rc = lbfgsb_hip_objective(ctx, int(kind, c_int), x, g, c_loc(f64))
         f = real(f64, wp)
      else
         rc = lbfgsb_hip_objective(ctx, int(kind, c_int), x, g, c_null_ptr)
      end if
      end subroutine lbfgsb_objective

      function lbfgsb_error_message() result(msg)
      character(len=:),allocatable :: msg
      type(c_ptr) :: p
      character(kind=c_char),pointer :: cs(:)
      integer :: k, i
      p = lbfgsb_hip_last_error()
      msg = ''
      if (.not. c_associated(p)) return
      k = int(c_strlen(p))
      if (k <= 0) return
      call c_f_pointer(p, cs, [k])
      allocate (character(len=k) :: msg)
      do i = 1, k
         msg(i:i) = cs(i)
      end do
      end function lbfgsb_error_message

      subroutine lbfgsb_release(Isave)
      integer :: Isave(44)
      integer(c_int32_t) :: rc
      rc = lbfgsb_hip_release_host_ik(Isave, int(ibytes, c_int32_t))
      end subroutine lbfgsb_release

      end module lbfgsb_module
