// solver_wide.inl -- member functions of Solver<T> (included inside the class body in solver.hip):
// the iteration for m > 32.  The reference puts no upper limit on m (src/lbfgsb.f90:93-97); the fused
// passes are unrolled for at most MAXM = 32 pairs.  A context with more pairs has every n-dimensional step
// as an unfused piece of k_wide.hip, in tiles of <= 32 logical columns (the circular column addressing makes a
// tile just another (head, col) pair): matupd's dot products, cauchy's p = W'd, formk's inner products, cmprlb's
// r, subsm's W'r and Newton direction, the projected step -- the reference's own sequence of steps, element-wise
// arithmetic in its operation order.  By default three pieces of the fused route run in front of them (DESIGN.md
// 4f): the update pass, split over the columns (matupd's, cauchy's and formk's sums in one pass over W:
// wide_fused()), WN1 kept incrementally (wide_formk_incr), and W'Z r in closed form with cmprlb's and subsm's
// updates of r as one axpy pass (wide_subspace, closed) -- two passes over W per iteration here too.
bool wide() const { return m > lbk::MAXM; }

// outY[j] = Wy_j' v, outS[j] = Ws_j' v for the col logical columns, tile by tile (one fetch per tile)
int wide_wtv(const T *v, int col, int head, double *outY, double *outS) {
  for (int j0 = 0; j0 < col; j0 += lbk::MAXM) {
    const int tc = std::min(lbk::MAXM, col - j0);
    const int h = (head - 1 + j0) % m + 1;
    lbk::launch_wtv<T>(q, n, W(), h, tc, v);
    const int MC = lbk::maxc_for(tc);
    CHK(fetch(2 * MC, 0, 0));
    for (int j = 0; j < tc; ++j) outY[j0 + j] = h_res[j], outS[j0 + j] = h_res[MC + j];
  }
  return 0;
}
// out_i += sum_j (Wy(i,j) a_j) / div + Ws(i,j) b_j, j ascending over all col columns
int wide_axpy(T *out, const double *a, const double *b, int col, int head, double div, int masked) {
  for (int j0 = 0; j0 < col; j0 += lbk::MAXM) {
    const int tc = std::min(lbk::MAXM, col - j0);
    const int h = (head - 1 + j0) % m + 1;
    lbk::Coef cf;
    std::memset(&cf, 0, sizeof cf);
    for (int j = 0; j < tc; ++j) cf.a[j] = a[j0 + j], cf.a[lbk::MAXM + j] = b[j0 + j];
    lbk::launch_tile_axpy<T>(q, n, W(), h, tc, cf, div, iwhere, masked, out);
  }
  return 0;
}
T *wcol(T *base, int head, int j) const { return base + (int64_t)((head - 1 + j) % m) * ld; }

// mainlb :812-824 + matupd :2291-2346: the new pair goes into its W slot at once; Sy's new row, Ss's new
// column and rr = y'y from W' s and W' y
int wide_matupd(const T *g, double stp, int head, int col, std::vector<double> &sy_row,
                std::vector<double> &ss_col, double &rr) {
  lbk::launch_pair_commit<T>(q, n, g, r, d, lbk::Pend{1, stp, 0}, W(), head, col);
  std::vector<double> oy(col), os(col);
  CHK(wide_wtv(wcol(ws, head, col - 1), col, head, oy.data(), os.data()));  // v = s
  sy_row.assign(oy.begin(), oy.end());  // s'Wy_j  (:2335)
  ss_col.assign(os.begin(), os.end());  // Ws_j's  (:2336)
  CHK(wide_wtv(wcol(wy, head, col - 1), col, head, oy.data(), os.data()));  // v = y
  rr = oy[col - 1];                     // y'y (:816)
  return 0;
}

// cauchy's n-loop (:1270-1330) without the fused p: the col = 0 scan, then p = W'd with d as a vector
int wide_cauchy_scan(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, int head, int col) {
  lbk::launch_cauchy_scan<T>(q, n, x, l, u, nbd, g, iwhere, tbrk, W(), head, 0);
  iw_dirty += 1.0;
  tbrk_valid = true;
  CHK(fetch(4, 1, 0));
  scan.f1 = h_res[0], scan.nbreak = h_res[1], scan.nunb = h_res[2], scan.nunbnz = h_res[3];
  scan.bkmin = h_res[4];
  if (col > 0) {
    lbk::launch_cauchy_dvec<T>(q, n, g, tbrk, xp);
    CHK(wide_wtv(xp, col, head, &scan.p[0], &scan.p[col]));
  }
  return 0;
}

// formk's inner products from scratch (:1756-1851), one masked column at a time:
//   Z Wy_j  -> Wy_i'(.) = Y'ZZ'Y(i,j),  Ws_i'(.) = R_z(i,j) (i <= j)
//   A Ws_j  -> Ws_i'(.) = S'AA'S(i,j),  Wy_i'(.) = L_a(j,i) (j > i)
int wide_formk(int col, int head) {
  lbh::Mat WN1{snd.data(), 2 * m};
  std::vector<double> a(col), b(col);
  for (int j = 0; j < col; ++j) {
    lbk::launch_masked_copy<T>(q, n, wcol(wy, head, j), iwhere, 1, xp);
    CHK(wide_wtv(xp, col, head, a.data(), b.data()));
    for (int i = j; i < col; ++i) WN1(i, j) = a[i];
    for (int i = 0; i <= j; ++i) WN1(m + i, j) = b[i];
    lbk::launch_masked_copy<T>(q, n, wcol(ws, head, j), iwhere, 0, xp);
    CHK(wide_wtv(xp, col, head, a.data(), b.data()));
    for (int i = j; i < col; ++i) WN1(m + i, m + j) = b[i];
    for (int i = 0; i < j; ++i) WN1(m + j, i) = a[i];
  }
  return 0;
}

// ... and incrementally, the way the reference keeps WN1 (:1735-1851), for the usual iteration: only the new
// pair's row and column are missing -- the free part of the new Wy column and the active part of the new Ws
// column against all columns: two masked copies + two tiled W'v instead of 2 col of them (at m = 40, n = 2e7
// formk from scratch was 170 of the iteration's 184 ms) -- and the rows that entered or left the free set
// change the old entries by their own outer products (:1801-1851): their records are gathered
// (rows_gather_kernel; the sparse patch kernel of the fused route keeps its tiles in LDS and is sized for 32
// pairs), summed on the host in the order of the sorted list, and reduced over the ranks like every sum.
// More changed rows than WIDE_PATCH_DOUBLES of records: from scratch.
static constexpr int64_t WIDE_PATCH_DOUBLES = 1 << 22;
double *wide_rows = nullptr;
size_t wide_rows_cap = 0;
int wide_formk_incr(int col, int head, bool updatd, int iupdat, bool &done, const double *nr_pass = nullptr) {
  done = false;
  const int upcl = updatd ? col - 1 : col;
  const int64_t nchg = nenter_g + (nglob + 1 - ileave_g);
  std::vector<double> P;
  if (nchg > 0 && upcl > 0) {
    if (nchg > (int64_t)CHG_CAP || nchg * 2 * upcl > WIDE_PATCH_DOUBLES) return 0;  // (the same on every rank)
    CHK(commit_pending((const T *)cg, col, head));
    const uint32_t nl = std::min<uint32_t>(chg_local, CHG_CAP);
    const int tri = upcl * (upcl + 1) / 2, E = 2 * upcl * upcl + upcl;
    P.assign((size_t)E, 0.0);
    if (nl > 0) {
      const size_t need = (size_t)nl * 2 * upcl;
      if (need > wide_rows_cap) {
        if (wide_rows) (void)hipFree(wide_rows);
        wide_rows = nullptr, wide_rows_cap = 0;
        HIPCHK(hipMalloc(&wide_rows, need * sizeof(double)));
        wide_rows_cap = need;
      }
      const uint32_t *lst = lbk::launch_sort_u32(q, sort_tmp, sort_tmp_bytes, d_chg, idx[1], nl);
      lbk::launch_rows_gather<T>(q, lst, nl, W(), head, upcl, wide_rows);
      std::vector<double> rows(need);
      std::vector<uint32_t> ids(nl);
      HIPCHK(hipMemcpyAsync(rows.data(), wide_rows, need * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipMemcpyAsync(ids.data(), lst, (size_t)nl * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      nsync++;
      for (uint32_t k = 0; k < nl; ++k) {  // (+ entering, - leaving: the flag in bit 31)
        const double sg = (ids[k] & 0x80000000u) ? -1.0 : 1.0;
        const double *y = &rows[(size_t)k * 2 * upcl], *s_ = y + upcl;
        for (int i = 0; i < upcl; ++i) {
          const double yi = sg * y[i], si = sg * s_[i];
          double *py = &P[(size_t)i * (i + 1) / 2], *ps = &P[(size_t)tri + (size_t)i * (i + 1) / 2];
          for (int j = 0; j <= i; ++j) py[j] += yi * y[j], ps[j] += si * s_[j];
          double *pm = &P[(size_t)2 * tri + (size_t)i * upcl];
          for (int j = 0; j < upcl; ++j) pm[j] += si * y[j];
        }
      }
    }
    if (nranks > 1) {  // every rank's share, summed in rank order like every other sum
      HIPCHK(hipMemcpyAsync(q.d_res, P.data(), (size_t)E * sizeof(double), hipMemcpyHostToDevice, stream));
      CHK(fetch(E, 0, 0));
      P.assign(h_res, h_res + E);
    }
  }
  std::vector<double> nr;
  if (updatd && nr_pass) {
    nr.assign(nr_pass, nr_pass + (size_t)4 * col);
  } else if (updatd) {
    CHK(commit_pending((const T *)cg, col, head));
    nr.assign((size_t)4 * col, 0.0);
    std::vector<double> a(col), b(col);
    lbk::launch_masked_copy<T>(q, n, wcol(wy, head, col - 1), iwhere, 1, xp);
    CHK(wide_wtv(xp, col, head, a.data(), b.data()));
    for (int j = 0; j < col; ++j) nr[0 * col + j] = a[j], nr[3 * col + j] = b[j];  // Y'ZZ'Y row, R_z column
    lbk::launch_masked_copy<T>(q, n, wcol(ws, head, col - 1), iwhere, 0, xp);
    CHK(wide_wtv(xp, col, head, a.data(), b.data()));
    for (int j = 0; j < col; ++j) nr[1 * col + j] = b[j], nr[2 * col + j] = a[j];  // S'AA'S row, L_a row
  }
  CHK(formk_incremental(col, head, updatd, iupdat, nr.data(), col, &P));
  done = true;
  return 0;
}

// cmprlb (:1548-1586) with r as a vector: tbrk = r on the free rows, 0 elsewhere
int wide_cmprlb(const T *x, const T *l, const T *u, const T *g, double theta, int col, int head, bool cnstnd,
                int &info) {
  const bool plain = !cnstnd && col > 0;
  std::vector<double> a1(col, 0.0), a2(col, 0.0);
  if (!plain) {
    if (lbh::bmv(m, sy.data(), wt.data(), col, &wa8m[2 * m], &wa8m[0]) != 0) {
      info = -8;
      return 0;
    }
    for (int j = 0; j < col; ++j) a1[j] = wa8m[j], a2[j] = theta * wa8m[col + j];  // :1576-1577
  }
  CHK(ensure_z(x, l, u, g));
  lbk::launch_cmprlb_init<T>(q, n, x, g, z, iwhere, theta, plain ? 1 : 0, tbrk);
  tbrk_valid = false;
  if (!plain) CHK(wide_axpy(tbrk, a1.data(), a2.data(), col, head, 1.0, 1));
  return 0;
}

// subsm (:2676-2885) with r (tbrk, from wide_cmprlb), W'r and the Newton direction as vectors.
// xp_first: the safeguard copy xp = xcp (:2787) is taken from z before the projected step overwrites
// it (contexts that mirror the reference's arrays, the routine doors); otherwise it is written from
// the functional form of this call's Cauchy point, and only if the backtracking branch needs it.
int wide_subsm(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, double theta, int col,
               int head, bool xp_first, int &iword, int &info) {
  const int ipr = quiet ? -1 : print_level;
  double *wv = &wa8m[0];
  {
    std::vector<double> oy(col), os(col);
    CHK(wide_wtv(tbrk, col, head, oy.data(), os.data()));  // :2742-2754 (r is 0 off the free rows)
    for (int i = 0; i < col; ++i) wv[i] = oy[i], wv[col + i] = theta * os[i];
  }
  nthreepass++;
  if (ipr >= 99) std::fprintf(rep.out, "\n----------------SUBSM entered-----------------\n\n");  // :2738
  lbh::Mat WN{wn.data(), 2 * m};
  const int col2 = 2 * col;
  info = lbh::dtrsl(WN, col2, wv, 11);
  if (info != 0) return 0;
  for (int i = 0; i < col; ++i) wv[i] = -wv[i];
  info = lbh::dtrsl(WN, col2, wv, 1);
  if (info != 0) return 0;
  CHK(wide_axpy(tbrk, wv, wv + col, col, head, theta, 1));  // :2770-2778
  return wide_subsm_tail(x, l, u, nbd, g, theta, xp_first, iword);
}
// ... from the Newton direction in tbrk on: the projected step, iword, the backtracking branch (:2780-2885)
int wide_subsm_tail(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, double theta, bool xp_first,
                    int &iword) {
  const int ipr = quiet ? -1 : print_level;
  if (xp_first) HIPCHK(hipMemcpyAsync(xp, z, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));  // :2787
  lbk::launch_subsm_project<T>(q, n, z, tbrk, x, g, l, u, nbd, iwhere, 1.0 / theta);  // :2780-2827
  z_valid = true;
  CHK(fetch(2, 0, 0));
  iword = h_res[0] > 0.0 ? 1 : 0;
  const double dd_p = h_res[1];
  ls.ready = false;
  if (iword == 0 || dd_p <= 0.0) {  // :2820, :2828
    if (ipr >= 99) std::fprintf(rep.out, "\n----------------exit SUBSM --------------------\n\n");  // :2883
    return 0;
  }
  if (rep.out && !quiet && print_level >= 0) {
    std::fprintf(rep.out, " Positive dir derivative in projection \n");
    std::fprintf(rep.out, " Using the backtracking step \n");
  }
  if (!xp_first) CHK(write_xcp(xp, x, l, u, g));
  lbk::launch_subsm_alpha<T>(q, n, xp, tbrk, l, u, nbd, iwhere);
  CHK(fetch(0, 1, 0));
  const double alpha = std::min(1.0, h_res[0]);
  int64_t ibd = -1;
  if (alpha < 1.0) {
    lbk::launch_subsm_argalpha<T>(q, n, row0, xp, tbrk, l, u, nbd, iwhere, alpha);
    CHK(fetch(0, 1, 0));
    ibd = (int64_t)h_res[0];
  }
  lbk::launch_subsm_backtrack<T>(q, n, row0, z, xp, tbrk, l, u, iwhere, alpha, ibd);
  if (ipr >= 99) std::fprintf(rep.out, "\n----------------exit SUBSM --------------------\n\n");
  return 0;
}

// what follows the fused r pass's four sums (in the same call, or -- deferred -- in the next one): the line-search
// set-up values, and for an uphill projected step (:2828) the iteration's r pass again through the unfused steps
struct WideLand {
  bool pending = false;
  std::vector<double> ca, cb;
  double theta = 1.0;
  int col = 0, head = 1;
  bool plain = false;
} wl;
int wide_unfused_r(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, int &iword) {
  CHK(ensure_z(x, l, u, g));
  lbk::launch_cmprlb_init<T>(q, n, x, g, z, iwhere, wl.theta, wl.plain ? 1 : 0, tbrk);
  tbrk_valid = false;
  CHK(wide_axpy(tbrk, wl.ca.data(), wl.cb.data(), wl.col, wl.head, 1.0, 1));
  return wide_subsm_tail(x, l, u, nbd, g, wl.theta, (flags & LBFGSB_F_MIRROR_INDEX) != 0, iword);
}
int wide_land(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, const double *R, int &iword) {
  wl.pending = false;
  iword = R[0] > 0.0 ? 1 : 0;
  const double dd_p = R[1];
  ls.ready = true, ls.x_is_z = ls_unit_step, ls.gd = dd_p, ls.dtd = R[2], ls.stpmx = R[3];
  if (iword == 0 || dd_p <= 0.0) return 0;  // :2820, :2828
  // the backtracking branch (:2830-2879): from the iterate again, through the unfused steps
  ls.ready = false, d_impl = z_in_x = false, z_valid = false;
  if (ls.x_is_z && !pp) HIPCHK(hipMemcpyAsync(xmut, t, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
  ls.x_is_z = false;
  return wide_unfused_r(x, l, u, nbd, g, iword);
}

// closed: W'Z r in closed form from the walk's p and WN1 (subspace_closed_form: no W'r pass), and -- with wv known
// before any row of r exists -- cmprlb's and subsm's two updates of r as ONE pass over W:
//   r = r0 + Wy (a1 + wv_y / theta) + Ws (a2 + wv_s)      (the reference adds W (M c) first, then divides the
// Wy product by theta: the same sums in another association; the caller has checked closed_form_safe)
int wide_subspace(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, double theta, int col,
                  int head, bool cnstnd, int &iword, int &info, bool closed = false) {
  if (closed) {
    const bool plain = !cnstnd && col > 0;
    std::vector<double> a1(col, 0.0), a2(col, 0.0), ca(col), cb(col);
    if (!plain) {
      if (lbh::bmv(m, sy.data(), wt.data(), col, &wa8m[2 * m], &wa8m[0]) != 0) {
        info = -8;
        return 0;
      }
      for (int j = 0; j < col; ++j) a1[j] = wa8m[j], a2[j] = theta * wa8m[col + j];  // :1576-1577
    }
    double *wv = &wa8m[0];
    subspace_closed_form(col, theta, a1.data(), a2.data(), wv);
    nclosed++;
    if (print_level >= 99 && !quiet) std::fprintf(rep.out, "\n----------------SUBSM entered-----------------\n\n");
    lbh::Mat WN{wn.data(), 2 * m};
    const int col2 = 2 * col;
    info = lbh::dtrsl(WN, col2, wv, 11);
    if (info != 0) return 0;
    for (int i = 0; i < col; ++i) wv[i] = -wv[i];
    info = lbh::dtrsl(WN, col2, wv, 1);
    if (info != 0) return 0;
    for (int j = 0; j < col; ++j) ca[j] = a1[j] + wv[j] / theta, cb[j] = a2[j] + wv[col + j];
    if (wide_tail_on && !(flags & LBFGSB_F_MIRROR_INDEX) && print_level < 99) {
      // The r pass with cmprlb's start and subsm's tail folded into its first and last tile (k_wide.hip,
      // tile_axpy_fused_kernel): neither xcp nor r0 is written, the Newton direction never leaves the last tile,
      // and the projected step, d = z - x, the line-search set-up sums and the first trial point come out of that
      // tile as they come out of subsm_update_kernel for m <= 32 -- six vector kernels and two host syncs less per
      // iteration (m = 48, n = 2e7: 6.86 -> 6.25 ms).  Uphill projected steps (:2828) redo the unfused sequence below.
      const bool lean = lean_on && ls_unit_step && (cnstnd || two_pass);
      bool deferred = false;
      lbk::WideTail<T> tail{x, g, l, u, nbd, gcp.tsum, theta, plain ? 1 : 0, ls_do_stpmx ? 1 : 0,
                          lean ? (T *)nullptr : z, lean ? (T *)nullptr : d, pp ? (T *)nullptr : t,
                          pp ? (T *)nullptr : r, ls_unit_step ? xmut : (T *)nullptr};
      if (gcp.copy_x) tail.tsum = 0.0;  // (xcp = x: no walk behind this Cauchy point)
      if (wide_one_on && col <= lbk::WIDE_MAXC) {
        // all tiles in one launch, the pending pair committed by it (m = 48, n = 2e7: dz_materialise, pair_commit
        // and two tile launches, 3.3 ms, become one launch of 2.7 ms: 6.25 -> 5.76 ms per iteration)
        lbk::CoefWide cw;
        std::memset(&cw, 0, sizeof cw);
        for (int j = 0; j < col; ++j) cw.a[j] = ca[j], cw.a[lbk::WIDE_MAXC + j] = cb[j];
        tail.l = lk(l), tail.u = uk(u);
        // LBFGSB_F_DEFER_LNSRCH as for m <= 32 (subspace()): the four sums are not waited for, they come over with
        // the fetch of the next call's first pass and wide_land runs there
        deferred = defer_on && ls_unit_step && !(flags & LBFGSB_F_PARALLEL_GCP);
        q.res_off = deferred ? DEFER_OFF : 0;
        if (deferred && fold_fin) q.part_sel = 2, q.hold_fin = true;
        // (the host stretch between the landing of the trial point's sums and this launch: as subspace() counts it)
        seg(4);
        if (t_mid0 > 0.0) t_mid += now_s() - t_mid0, n_mid++, t_mid0 = 0.0;
        lbk::launch_wide_r_pass<T>(q, n, W(), head, col, cw, iwhere, nbk(), ub_mask, tail, pend, r, d_src());
        q.res_off = 0, q.part_sel = 0;
        pend.on = 0, pend.impl = 0;  // the pass stored the pair into its W slot
      } else {
        CHK(commit_pending(g, col, head));
      for (int j0 = 0; j0 < col; j0 += lbk::MAXM) {
        const int tc = std::min(lbk::MAXM, col - j0);
        lbk::Coef cf;
        std::memset(&cf, 0, sizeof cf);
        for (int j = 0; j < tc; ++j) cf.a[j] = ca[j0 + j], cf.a[lbk::MAXM + j] = cb[j0 + j];
        lbk::launch_tile_axpy_fused<T>(q, n, W(), (head - 1 + j0) % m + 1, tc, cf, iwhere, tbrk, j0 == 0 ? 1 : 0,
                                       j0 + lbk::MAXM >= col ? 1 : 0, tail);
      }
      }
      tbrk_valid = false;
      d_impl = z_in_x = lean;
      z_valid = !lean;
      if (lean) x_lean = xmut;
      if (pp) t = const_cast<T *>(x), r = const_cast<T *>(g);  // t = x, r = g (:2235-2236) as a change of roles
      wl.pending = true, wl.ca = ca, wl.cb = cb, wl.theta = theta, wl.col = col, wl.head = head, wl.plain = plain;
      if (deferred) {
        defer_live = true, ls.deferred = true;
        ndeferred++;
        return 0;
      }
      CHK(fetch(3, 1, 0));
      return wide_land(x, l, u, nbd, g, h_res, iword);
    }
    CHK(commit_pending(g, col, head));  // (the unfused steps read the newest pair from W)
    wl.ca = ca, wl.cb = cb, wl.theta = theta, wl.col = col, wl.head = head, wl.plain = plain;
    return wide_unfused_r(x, l, u, nbd, g, iword);
  }
  CHK(commit_pending(g, col, head));  // (the unfused steps read the newest pair from W)
  CHK(wide_cmprlb(x, l, u, g, theta, col, head, cnstnd, info));
  if (info != 0) return 0;
  return wide_subsm(x, l, u, nbd, g, theta, col, head, (flags & LBFGSB_F_MIRROR_INDEX) != 0, iword, info);
}
