!  ref_cabi.f90 -- ORACLE SUPPORT (test infrastructure only, NOT the product).
!
!  A bind(C) doorway into the *real* reference `setulb`
!  (/root/reference/src/lbfgsb.f90:88) so that tests, the golden-vector
!  generator and bench.py's cpu_baseline leg can call the untouched reference
!  through ctypes.  This file is ours; the four reference sources are compiled
!  where they lie by oracle/Makefile and only the resulting objects/.so land in
!  oracle/_ref/ (git-ignored).
!
!  Integer/logical width follows the build: default 4-byte, or 8-byte when
!  compiled with -fdefault-integer-8 (needed for n = 1e8, SURVEY.md section 0).
!  The C caller must pass arrays of that width.
module ref_cabi
   use iso_c_binding
   use lbfgsb_module, only: setulb, wp => lbfgsp_wp
   implicit none
   integer, parameter :: ik = kind(0)
contains

   subroutine ref_setulb(n, m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, task, &
                         iprint, csave, lsave, isave, dsave) bind(C, name='ref_setulb')
      integer(ik), value :: n, m, iprint
      real(wp) :: x(n), l(n), u(n), g(n), f
      real(wp), value :: factr, pgtol
      real(wp) :: wa(*), dsave(29)
      integer(ik) :: nbd(n), iwa(*), lsave(4), isave(44)
      character(kind=c_char) :: task(60), csave(60)

      character(len=60) :: ftask, fcsave
      logical :: flsave(4)
      integer :: i

      do i = 1, 60
         ftask(i:i) = task(i)
         fcsave(i:i) = csave(i)
      end do
      do i = 1, 4
         flsave(i) = lsave(i) /= 0
      end do

      call setulb(n, m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, ftask, iprint, &
                  fcsave, flsave, isave, dsave)

      do i = 1, 60
         task(i) = ftask(i:i)
         csave(i) = fcsave(i:i)
      end do
      do i = 1, 4
         if (flsave(i)) then
            lsave(i) = 1
         else
            lsave(i) = 0
         end if
      end do
   end subroutine ref_setulb

   ! size in bytes of the build's integer and real, so a ctypes caller can
   ! check it is talking to the variant it thinks it is.
   subroutine ref_kinds(int_bytes, real_bytes) bind(C, name='ref_kinds')
      integer(c_int), intent(out) :: int_bytes, real_bytes
      int_bytes = int(storage_size(0_ik)/8, c_int)
      real_bytes = int(storage_size(1.0_wp)/8, c_int)
   end subroutine ref_kinds

end module ref_cabi
