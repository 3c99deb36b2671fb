// solver_doors.inl -- member functions of Solver<T> (included inside the class body in solver.hip):
// the ROUTINE doors of the C ABI.  Each runs ONE routine of the reference on the
// state of this context -- W, Sy, Ss, Wt, WN, WN1, z, r, d, t, xp, the 8m work vectors, iwhere, Index,
// Indx2: exactly what lbfgsb_hip_import_state loads and lbfgsb_hip_export_state reads back, in the
// reference's wa / iwa layout -- so that every routine can be compared with its CPU twin on the same
// inputs.  They drive the library's own code: cauchy() is the function every iteration calls, freev and
// matupd share their halves with the iteration's phases, cmprlb / subsm / matupd's n-length sums run
// through the unfused tile functions of solver_wide.inl (the m > 32 path, valid for every m), formk is
// the from-scratch Gram pass + the host factorisations.  The fused passes of the hot path have no
// routine-shaped door: they are covered call by call (tests/test_gpu_parity.py, one-step parity).
//
// Single rank only (a routine's arguments are the reference's: whole vectors); a door leaves the
// context fit for further doors and for import_state, not for continuing a setulb run in the middle.
  int door_ready(const int32_t *nbd) {
    if (nranks != 1) return fail(LBFGSB_E_STATE, "routine doors are single-rank");
    HIPCHK(hipSetDevice(device));
    spec.valid = false, spcand.valid = false, scan.ready = false, pend.on = 0, pend.impl = 0;
    d_impl = false, z_in_x = false, pre_valid = false, closed_ok = false, nrpre.valid = false;
    ls.ready = false, ls.x_is_z = false;
    t = t_own, r = r_own;
    if (nbd) {
      // (plain nbd bytes, streamed bounds: a run that ended in dictionary / uniform mode must not have its codes or
      //  constants applied to the arrays a door is handed, as import_state does)
      nbd8_src = nullptr;
      ub_mask = 0;
      CHK(ensure_nbd8(nbd));
    }
    return 0;
  }
  static double eps_T() {
    return sizeof(T) == 4 ? (double)std::numeric_limits<float>::epsilon() : std::numeric_limits<double>::epsilon();
  }

  // level-1 BLAS, src/lbfgsb_blas_module.F90:37-277 at its n-length call sites (d = z - x :720-722, y = g - r
  // :812-816, s = stp d :822, the dots of :816 / :2196 / :2244 / :2335): plain vectors, no state of the context
  // involved, any number of ranks -- the dot is reduced over the ranks like every other sum (one all-gather)
  int r_vec_sub(const void *a, const void *b, void *out) override {
    HIPCHK(hipSetDevice(device));
    lbk::launch_vec_sub<T>(q, n, (const T *)a, (const T *)b, (T *)out);
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  int r_vec_scale(double alpha, void *v) override {
    HIPCHK(hipSetDevice(device));
    lbk::launch_vec_scale<T>(q, n, (double)(T)alpha, (T *)v);  // (REAL32: the factor as the reference holds it)
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  int r_dot(const void *a, const void *b, double *out) override {
    HIPCHK(hipSetDevice(device));
    lbk::launch_dot<T>(q, n, (const T *)a, (const T *)b);
    CHK(fetch(1, 0, 0));
    *out = h_res[0];
    return 0;
  }

  // active :965-1040 -- x projected onto the box in place, iwhere initialised; out3 = prjctd, cnstnd, boxed
  int r_active(void *x, const void *l, const void *u, const int32_t *nbd, int32_t *out3) override {
    CHK(door_ready(nbd));
    lbk::launch_active<T>(q, n, (T *)x, (const T *)l, (const T *)u, nbd, iwhere, wasfree);
    iw_dirty = 1.0;
    index_valid = false;
    CHK(fetch(4, 0, 0));
    out3[0] = h_res[0] > 0.0, out3[1] = h_res[1] > 0.0, out3[2] = h_res[2] == 0.0;
    if (prevfree) HIPCHK(hipMemsetAsync(prevfree, 1, (size_t)n, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  // errclb :1601-1643 -- task is left alone if the input is fine, else 'ERROR: ...' with info / k
  int r_errclb(const void *l, const void *u, const int32_t *nbd, double factr, char *task, int32_t *info,
               int64_t *k) override {
    CHK(door_ready(nullptr));
    *info = 0, *k = 0;
    if (nglob <= 0) lbh::str60_set(task, "ERROR: N <= 0");
    if (m <= 0) lbh::str60_set(task, "ERROR: M <= 0");
    if (factr < 0.0) lbh::str60_set(task, "ERROR: FACTR < 0");
    lbk::launch_errclb<T>(q, n, row0, (const T *)l, (const T *)u, nbd);
    CHK(fetch(0, 0, 5));
    const int64_t k6 = (int64_t)h_res[0], k7 = (int64_t)h_res[1];
    if (k6 > 0 || k7 > 0) {  // (the reference's loop keeps the LAST offender: the larger index wins)
      if (k6 > k7) {
        lbh::str60_set(task, "ERROR: INVALID NBD");
        *info = -6, *k = k6;
      } else {
        lbh::str60_set(task, "ERROR: NO FEASIBLE SOLUTION");
        *info = -7, *k = k7;
      }
    }
    return 0;
  }

  // cauchy :1157-1532 -- reads W, Sy, Wt, iwhere of the state; leaves xcp in z (and in xcp_out, a device
  // vector, if given), iwhere, p / c / wbp / v in the 8m work vectors, nseg, info
  int r_cauchy(const void *x, const void *l, const void *u, const int32_t *nbd, const void *g, double theta,
               int col, int head, double sbgnrm, void *xcp_out, int32_t *nseg, int32_t *info) override {
    if (col < 0 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "cauchy: bad col/head");
    CHK(door_ready(nbd));
    int ns = 0, inf = 0;
    CHK(cauchy((const T *)x, (const T *)l, (const T *)u, nbd, (const T *)g, theta, col, head, sbgnrm, eps_T(),
               ns, inf));
    if (inf == 0) {
      CHK(ensure_z((const T *)x, (const T *)l, (const T *)u, (const T *)g));
      if (xcp_out) HIPCHK(hipMemcpyAsync(xcp_out, z, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));
    *nseg = ns, *info = inf;
    return 0;
  }

  // freev :1980-2059 -- from iwhere and the free set of the previous iteration (Index(1:nfree) of the
  // imported state): the counts, wrk, and -- in contexts that mirror them -- Index and Indx2
  int r_freev(int iter, int cnstnd, int updatd, int64_t *nfree, int64_t *nenter, int64_t *ileave,
              int32_t *wrk) override {
    CHK(door_ready(nullptr));
    const bool track = iter > 0 && cnstnd;
    CHK(freev_launch(track));
    CHK(fetch(4, 0, 0));
    *wrk = freev_land(track, updatd != 0) ? 1 : 0;
    *nfree = nfree_g, *nenter = nenter_g, *ileave = ileave_g;
    if (index) lbk::launch_freev_lists(q, n, iwhere, prevfree, track ? 1 : 0, index, indx2, scan_tmp);
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  // formk :1681-1908 -- WN1's inner products from scratch over the free / active rows of iwhere (the
  // reference keeps them incrementally: same sums, other order), then the assembly of WN and its two
  // Cholesky factorisations on the host; WN1 and WN of the state are replaced
  int r_formk(int col, int head, double theta, int32_t *info) override {
    if (col < 1 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "formk: bad col/head");
    CHK(door_ready(nullptr));
    int inf = 0;
    CHK(formk(col, head, theta, inf));
    *info = inf;
    return 0;
  }

  // cmprlb :1548-1586 -- r = -Z'(B(xcp - x) + g) from z (= xcp), W, Sy, Wt, c (work vector 2) and iwhere;
  // r_out (device, n values): r scattered to its rows, 0 on the rows that are not free
  int r_cmprlb(const void *x, const void *g, double theta, int col, int head, int cnstnd, void *r_out,
               int32_t *info) override {
    if (col < 0 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "cmprlb: bad col/head");
    CHK(door_ready(nullptr));
    if (!z_valid) return fail(LBFGSB_E_STATE, "cmprlb: no Cauchy point in the state (cauchy or import_state first)");
    int inf = 0;
    CHK(wide_cmprlb((const T *)x, nullptr, nullptr, (const T *)g, theta, col, head, cnstnd != 0, inf));
    if (inf == 0 && r_out)
      HIPCHK(hipMemcpyAsync(r_out, tbrk, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    *info = inf;
    return 0;
  }

  // subsm :2676-2885 -- r_in (device, n values, as r_cmprlb writes it), WN of the state (formk), z = xcp
  // in, z = the subspace minimiser out (also in xhat_out, device, if given); xp = xcp kept (:2787)
  int r_subsm(const void *x, const void *l, const void *u, const int32_t *nbd, const void *g, const void *r_in,
              double theta, int col, int head, void *xhat_out, int32_t *iword, int32_t *info) override {
    if (col < 1 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "subsm: bad col/head");
    CHK(door_ready(nbd));
    if (!z_valid) return fail(LBFGSB_E_STATE, "subsm: no Cauchy point in the state (cauchy or import_state first)");
    if (r_in != (const void *)tbrk)
      HIPCHK(hipMemcpyAsync(tbrk, r_in, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    tbrk_valid = false;
    int iw = 0, inf = 0;
    CHK(wide_subsm((const T *)x, (const T *)l, (const T *)u, nbd, (const T *)g, theta, col, head, true, iw, inf));
    if (inf == 0 && xhat_out)
      HIPCHK(hipMemcpyAsync(xhat_out, z, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    *iword = iw, *info = inf;
    return 0;
  }

  // lnsrlb :2174-2275 (with mainlb's d = z - x, :720-722, on the first call of an iteration's search):
  // sc = fold, gd, gdold, stp, dnorm, dtd, xstep, stpmx; ic = iter, ifun, iback, nfgv, info, boxed, cnstnd;
  // task / csave / isave2(2) / dsave13(13) as the reference's.  z, d, t, r are the state's.
  int r_lnsrlb(void *x, const void *l, const void *u, const int32_t *nbd, const void *g, double f, double *sc,
               int32_t *ic, char *task, char *csave, int32_t *isave2, double *dsave13) override {
    CHK(door_ready(nbd));
    const double big = 1.0e10, ftol = 1.0e-3, gtol = 0.9, xtol = 0.1;
    double &fold = sc[0], &gd = sc[1], &gdold = sc[2], &stp = sc[3], &dnorm = sc[4], &dtd = sc[5],
           &xstep = sc[6], &stpmx = sc[7];
    const int iter = ic[0];
    int32_t &ifun = ic[1], &iback = ic[2], &nfgv = ic[3], &info = ic[4];
    const bool boxed = ic[5] != 0, cnstnd = ic[6] != 0;
    if (!lbh::str60_pre(task, "FG_LN")) {
      if (!z_valid) return fail(LBFGSB_E_STATE, "lnsrlb: no z in the state");
      lbk::launch_lnsrlb_begin<T>(q, n, z, (const T *)x, (const T *)g, (const T *)l, (const T *)u, nbd, d, t, r,
                                  (cnstnd && iter != 0) ? 1 : 0);
      CHK(fetch(2, 1, 0));
      dtd = h_res[0], gd = h_res[1];
      dnorm = std::sqrt(dtd);
      stpmx = big;
      if (cnstnd) stpmx = iter == 0 ? 1.0 : std::min(big, h_res[2]);
      stp = (iter == 0 && !boxed) ? std::min(1.0 / dnorm, stpmx) : 1.0;
      fold = f;
      ifun = 0, iback = 0;
      lbh::str60_set(csave, "START");
    } else {
      lbk::launch_lnsrlb_eval<T>(q, n, (const T *)x, (const T *)l, (const T *)u, nbd, (const T *)g, d);
      CHK(fetch(1, 0, 1));
      gd = h_res[0];
    }
    if (ifun == 0) {
      gdold = gd;
      if (gd >= 0.0) {  // :2247-2253
        info = -4;
        return 0;
      }
    }
    lbh::dcsrch(f, gd, stp, ftol, gtol, xtol, 0.0, stpmx, csave, isave2, dsave13);
    xstep = stp * dnorm;
    if (!lbh::str60_pre(csave, "CONV") && !lbh::str60_pre(csave, "WARN")) {
      lbh::str60_set(task, "FG_LNSRCH");
      ifun++, nfgv++;
      iback = ifun - 1;
      lbk::launch_lnsrlb_step<T>(q, n, (T *)x, z, d, t, stp);
    } else {
      lbh::str60_set(task, "NEW_X");
    }
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  // mainlb :812-824 + matupd :2291-2346 -- y = g - r (r of the state = the gradient the iteration started
  // from), s = stp * d, stored in W; Sy, Ss of the state updated; ip = iupdat (already counted up, as
  // mainlb :836 does before the call), col, head, itail in / out; theta = y'y / dr out.  dr and dtd are
  // the caller's, as the reference's matupd takes them.
  int r_matupd(const void *g, double stp, double dr, double dtd, int32_t *ip, double *theta_out) override {
    CHK(door_ready(nullptr));
    const int iupdat = ip[0];
    int col = ip[1], head = ip[2], itail = ip[3];
    if (iupdat < 1 || col < 0 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "matupd: bad pointers");
    if (iupdat <= m) {  // :2308-2315
      col = iupdat;
      itail = (head + iupdat - 2) % m + 1;
    } else {
      itail = itail % m + 1;
      head = head % m + 1;
    }
    double rr = 0.0;
    std::vector<double> wsy, wss;
    CHK(wide_matupd((const T *)g, stp, head, col, wsy, wss, rr));
    matupd_small(col, iupdat, wsy.data(), wss.data(), stp == 1.0 ? dtd : stp * stp * dtd, dr);
    HIPCHK(hipStreamSynchronize(stream));
    ip[1] = col, ip[2] = head, ip[3] = itail;
    *theta_out = rr / dr;
    return 0;
  }
