!  driver_dev -- the loop of the reference's test/driver2.f90:66-195 (customised stopping test, one line per
!  iterate) with everything n-long resident in HBM: x, l, u, nbd, g are device buffers obtained from hipMalloc
!  through iso_c_binding, the objective is the library's built-in separable bounded quadratic (BASELINE.md
!  section 3: the workload bench.py times), and the solver is driven through the device-pointer procedures of
!  lbfgsb_module (setulb_dev_pp / lbfgsb_objective: extensions beside the unchanged setulb).  Nothing n-long
!  crosses PCIe after the set-up: this is the Fortran caller that reaches the throughput bench.py reports.
!
!     driver_dev [n [m [iterations [warmup [mode]]]]]
!        n           rows (default 1000000)         m     pairs (default 10)
!        iterations  stop after this many (default 32)
!        warmup      iterations before the clock starts (default 12; the rest are timed)
!        mode        pp (default: ping-pong entry + LBFGSB_F_DEFER_LNSRCH + deferred f, as bench.py) | classic
!                    (setulb_dev, f fetched at every evaluation: an ordinary reverse-communication caller)
!
!  Output: driver2's "Iterate ... nfg = ... f = ... |proj g| = ..." lines, extended by nseg, nfree and the bit
!  patterns of f and |proj g| (Z16), then one "RATE" line.
      program driver_dev

      use lbfgsb_module, wp => lbfgsp_wp
      use iso_c_binding
      use iso_fortran_env, only: output_unit, int64

      implicit none

      interface
         function hipMalloc(ptr, nbytes) bind(C, name='hipMalloc') result(rc)
            import :: c_ptr, c_size_t, c_int
            type(c_ptr) :: ptr
            integer(c_size_t), value :: nbytes
            integer(c_int) :: rc
         end function hipMalloc
         function hipFree(ptr) bind(C, name='hipFree') result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: ptr
            integer(c_int) :: rc
         end function hipFree
         function hipMemcpy(dst, src, nbytes, kind) bind(C, name='hipMemcpy') result(rc)
            import :: c_ptr, c_size_t, c_int
            type(c_ptr), value :: dst, src
            integer(c_size_t), value :: nbytes
            integer(c_int), value :: kind
            integer(c_int) :: rc
         end function hipMemcpy
         function hipMemset(dst, val, nbytes) bind(C, name='hipMemset') result(rc)
            import :: c_ptr, c_size_t, c_int
            type(c_ptr), value :: dst
            integer(c_int), value :: val
            integer(c_size_t), value :: nbytes
            integer(c_int) :: rc
         end function hipMemset
      end interface
      integer(c_int), parameter :: H2D = 1

      integer               :: n, m, maxit, warm, iprint, rc, cur, flags, k
      logical               :: pp, compact
      real(wp), parameter   :: factr = 0.0_wp, pgtol = 0.0_wp
      character(len=60)     :: task, csave
      character(len=32)     :: arg
      logical               :: lsave(4)
      integer               :: isave(44)
      real(wp)              :: f, dsave(29)
      type(c_ptr)           :: ctx, xs(0:1), gs(0:1), dl, du, dnbd
      real(wp), allocatable, target :: hbuf(:)
      integer(c_int32_t), allocatable, target :: hnbd(:)
      integer(c_size_t)     :: vbytes
      integer(int64)        :: c0, c1, crate
      integer               :: it0

      n = 1000000; m = 10; maxit = 32; warm = 12; pp = .true.; iprint = -1
      if (command_argument_count() >= 1) then
         call get_command_argument(1, arg); read (arg, *) n
      end if
      if (command_argument_count() >= 2) then
         call get_command_argument(2, arg); read (arg, *) m
      end if
      if (command_argument_count() >= 3) then
         call get_command_argument(3, arg); read (arg, *) maxit
      end if
      if (command_argument_count() >= 4) then
         call get_command_argument(4, arg); read (arg, *) warm
      end if
      if (command_argument_count() >= 5) then
         call get_command_argument(5, arg); pp = trim(arg) /= 'classic'
      end if
      compact = .false.
      if (command_argument_count() >= 6) then
         call get_command_argument(6, arg); compact = trim(arg) == 'compact'
      end if

      ! ---- device buffers (16 bytes of slack: the library asks for 16-byte aligned pointers, hipMalloc gives 256)
      vbytes = int(n, c_size_t)*int(storage_size(1.0_wp)/8, c_size_t)
      do k = 0, 1
         call chk(hipMalloc(xs(k), vbytes), 'hipMalloc x')
         call chk(hipMalloc(gs(k), vbytes), 'hipMalloc g')
         call chk(hipMemset(xs(k), 0_c_int, vbytes), 'hipMemset x')      ! x0 = 0
         call chk(hipMemset(gs(k), 0_c_int, vbytes), 'hipMemset g')
      end do
      call chk(hipMalloc(dl, vbytes), 'hipMalloc l')
      call chk(hipMalloc(du, vbytes), 'hipMalloc u')
      call chk(hipMalloc(dnbd, int(n, c_size_t)*4_c_size_t), 'hipMalloc nbd')
      allocate (hbuf(n))
      hbuf = -1.0_wp                                                      ! l = -1
      call chk(hipMemcpy(dl, c_loc(hbuf), vbytes, H2D), 'hipMemcpy l')
      hbuf = 1.0_wp                                                       ! u = +1
      call chk(hipMemcpy(du, c_loc(hbuf), vbytes, H2D), 'hipMemcpy u')
      deallocate (hbuf)
      allocate (hnbd(n))
      hnbd = 2_c_int32_t                                                  ! both bounds
      call chk(hipMemcpy(dnbd, c_loc(hnbd), int(n, c_size_t)*4_c_size_t, H2D), 'hipMemcpy nbd')
      deallocate (hnbd)

      ! ---- the context replaces wa / iwa (src/lbfgsb.f90:250-284)
      flags = 0
      if (pp) flags = LBFGSB_F_NO_RETURN_SYNC + LBFGSB_F_DEFER_LNSRCH      ! the objective runs on the context's stream
      call lbfgsb_create(ctx, n, m, flags, rc)
      if (rc /= 0) then
         write (output_unit, '(2a)') ' lbfgsb_create failed: ', lbfgsb_error_message()
         error stop 1
      end if

      ! the two passes over W on the free-rows-first layout (DESIGN.md 4g), as bench.py runs them
      if (compact) then
         call lbfgsb_set_option(ctx, 'compact_w', 1.0d0, rc)
         if (rc /= 0) then
            write (output_unit, '(2a)') ' lbfgsb_set_option failed: ', lbfgsb_error_message()
            error stop 1
         end if
      end if

      task = 'START'
      cur = 0
      f = 0.0_wp
      it0 = -1
      call system_clock(count_rate=crate)
      c0 = 0; c1 = 0

      ! ------- the beginning of the loop (test/driver2.f90:104) ----------
      do while (task(1:2) == 'FG' .or. task == 'NEW_X' .or. task == 'START')

         if (pp) then
            call setulb_dev_pp(ctx, xs(0), xs(1), dl, du, dnbd, f, gs(0), gs(1), factr, pgtol, task, iprint, &
                               csave, lsave, isave, dsave, cur, rc)
         else
            call setulb_dev(ctx, xs(0), dl, du, dnbd, f, gs(0), factr, pgtol, task, iprint, csave, lsave, &
                            isave, dsave, rc)
         end if
         if (rc /= 0) then
            write (output_unit, '(2a)') ' setulb_dev failed: ', lbfgsb_error_message()
            error stop 1
         end if

         if (task(1:2) == 'FG') then
            ! f and g at x(cur) on the device; pp: the value rides with the next call's fetch
            if (pp) then
               call lbfgsb_objective(ctx, 0, xs(cur), gs(cur), rc)
            else
               call lbfgsb_objective(ctx, 0, xs(0), gs(0), rc, f)
            end if
            if (rc /= 0) error stop 2
         else if (task(1:5) == 'NEW_X') then
            if (isave(30) == warm) then
               call system_clock(c0); it0 = isave(30)
            end if
            if (isave(30) == maxit) call system_clock(c1)
            write (output_unit, '(2(a,i5,4x),a,1p,d12.5,4x,a,1p,d12.5,2(2x,a,i11),2(2x,z16.16))')            &
               'Iterate', isave(30), 'nfg =', isave(34), 'f =', f, '|proj g| =', dsave(13),                 &
               'nseg =', isave(33), 'nfree =', isave(38), transfer(real(f, c_double), 1_int64),             &
               transfer(real(dsave(13), c_double), 1_int64)
            if (isave(30) >= maxit) task = 'STOP: ITERATION LIMIT OF DRIVER_DEV'
         end if
      end do
      ! ---------- the end of the loop -------------

      if (task(1:4) /= 'STOP') write (output_unit, '(2a)') ' ended with task = ', trim(task)
      if (it0 >= 0 .and. c1 > c0) then
         write (output_unit, '(a,i0,a,i0,a,f12.4,a,f10.4,a)') 'RATE n=', n, ' iterations_timed=', maxit - it0, &
            ' iters_per_sec=', real(maxit - it0, c_double)*real(crate, c_double)/real(c1 - c0, c_double),      &
            ' ms_per_iter=', 1.0e3_c_double*real(c1 - c0, c_double)/real(crate, c_double)/real(maxit - it0, c_double), ''
      end if
      call lbfgsb_destroy(ctx)
      do k = 0, 1
         rc = hipFree(xs(k)); rc = hipFree(gs(k))
      end do
      rc = hipFree(dl); rc = hipFree(du); rc = hipFree(dnbd)

      contains

      subroutine chk(code, what)
      integer(c_int), intent(in) :: code
      character(len=*), intent(in) :: what
      if (code /= 0) then
         write (output_unit, '(3a,i0)') ' ', what, ' failed, hipError ', code
         error stop 3
      end if
      end subroutine chk

      end program driver_dev
