// solver_pgcp.inl -- member functions of Solver<T> (included inside the class body in solver.hip): the
// opt-in parallel generalized-Cauchy-point search for col > 0 (LBFGSB_F_PARALLEL_GCP; kernels in k_pgcp.hip)
  // ---- parallel GCP search for col > 0 (LBFGSB_F_PARALLEL_GCP; k_cauchy.hip "parallel GCP") ----
  double *pg_buf = nullptr;
  size_t pg_bytes = 0;
  void *pg_tmp = nullptr;
  size_t pg_tmp_bytes = 0;
  // all-gather of a large device buffer (count doubles per rank), rank-major into d_recv
  int allgather_big(const double *d_send, double *d_recv, size_t count) {
    ncoll++, coll_bytes += (int64_t)count * 8;
    if (comm) {
      if (g_rccl.AllGather(d_send, d_recv, count, ncclDouble, comm, stream) != ncclSuccess)
        return fail(LBFGSB_E_COMM, "ncclAllGather failed");
      return 0;
    }
    if (!cb_ag) return fail(LBFGSB_E_COMM, "multi-rank context without an all-gather");
    std::vector<double> hs(count), hr(count * (size_t)nranks);
    HIPCHK(hipMemcpyAsync(hs.data(), d_send, count * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (cb_ag(cb_user, hs.data(), hr.data(), (int64_t)(count * sizeof(double))) != 0)
      return fail(LBFGSB_E_COMM, "host all-gather callback failed");
    HIPCHK(hipMemcpyAsync(d_recv, hr.data(), hr.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));  // (hr is a temporary)
    return 0;
  }
  int parallel_gcp(const T *x, const T *l, const T *u, const T *g, double theta, int col, int head,
                   const double *p0, double *c, double f1_0, double f2_0, double f2_org, bool bnded,
                   int64_t nbreak, int &nseg, int &info, bool &done) {
    done = false;
    const int col2 = 2 * col;
    const bool multi = nranks > 1;
    // this rank's breakpoints in (t, index) order
    CHK(ensure_sel((size_t)n));
    uint32_t cnt = 0;
    CHK(local_count(-1.0, -1, std::numeric_limits<double>::max(), 0, cnt));  // (fills tbrk)
    // every rank's count (the ranks take the same decisions below)
    std::vector<double> counts(nranks, (double)cnt);
    if (multi) {
      CHK(put_header((double)cnt, 0.0));
      CHK(exchange(2));
      for (int rk = 0; rk < nranks; ++rk) counts[rk] = h_msg_all[2 * (size_t)rk];
    }
    int64_t nb = 0, nbmax = 0;
    for (double cv : counts) nb += (int64_t)cv, nbmax = std::max<int64_t>(nbmax, (int64_t)cv);
    const int64_t nbp = (nb + 31) / 32 * 32;          // stride of the arrays the scans run on
    const int64_t lbp = (nbmax + 31) / 32 * 32;       // stride of one rank's gathered arrays
    if (nb != nbreak || nb == 0) return 0;
    if (multi && (uint64_t)nranks * (uint64_t)lbp >= 0xffffffffull) return 0;
    const int narr_l = 4 + 2 * col2;                   // tt, dd, a0, gi, wb[col2], uu[col2]
    const size_t narr = 7 + 3 * (size_t)col2;          // + df2, a1, df1, sq[col2]
    const size_t small = (size_t)col2 * col2 + 4 * (size_t)col2 + 16 + 2 * (size_t)nranks + 96;
    const size_t gath = multi ? (size_t)narr_l * lbp * ((size_t)nranks + 1) : 0;
    const size_t bytes = (narr * (size_t)nbp + gath + small) * sizeof(double);
    bool fits = true;
    if (bytes > pg_bytes) {
      if (pg_buf) (void)hipFree(pg_buf);
      pg_buf = nullptr, pg_bytes = 0;
      size_t mfree = 0, mtotal = 0;
      (void)hipMemGetInfo(&mfree, &mtotal);
      if (bytes > mfree / 10 * 9 || hipMalloc(&pg_buf, bytes) != hipSuccess) {
        (void)hipGetLastError();
        fits = false;
      } else {
        pg_bytes = bytes;
      }
    }
    // (every allocation of this search happens BEFORE the ranks vote: a rank that cannot allocate
    //  votes "does not fit" and all of them replay the walk exactly -- none is left waiting in a
    //  collective)
    const size_t tb = std::max(lbk::scan_temp_bytes((size_t)nb), lbk::f2scan_temp_bytes((size_t)nb)) + 256;
    if (fits && tb > pg_tmp_bytes) {
      if (pg_tmp) (void)hipFree(pg_tmp);
      pg_tmp = nullptr, pg_tmp_bytes = 0;
      if (hipMalloc(&pg_tmp, tb) != hipSuccess) {
        (void)hipGetLastError();
        pg_tmp = nullptr;
        fits = false;
      } else {
        pg_tmp_bytes = tb;
      }
    }
    if (fits && ensure_sel(std::max((size_t)n, (size_t)nranks * (size_t)lbp)) != 0) {
      (void)hipGetLastError();
      fits = false;
      // (the window buffers of the exact replay must exist again)
      if (sel_alloc == 0) CHK(ensure_sel(SEL_CAP));
    }
    if (multi) {  // one rank short of memory sends every rank back to the exact replay
      CHK(put_header(fits ? 1.0 : 0.0, 0.0));
      CHK(exchange(2));
      for (int rk = 0; rk < nranks; ++rk) fits = fits && h_msg_all[2 * (size_t)rk] > 0.0;
    }
    if (!fits) return 0;
    nfullsort++;
    lbk::launch_cauchy_allkeys<T>(q, n, row0, tbrk, -1.0, -1, keys[0], idx[0]);
    lbk::launch_sort_pairs(q, sort_tmp, sort_tmp_bytes, keys[0], keys[1], idx[0], idx[1], (size_t)n);
    // arrays the scans run on (stride nbp); dd and a0 are dead after pgcp_terms and then hold the
    // 2 nb doubles of the f2 maps
    double *tt = pg_buf, *dd = tt + nbp, *a0 = dd + nbp, *df2 = a0 + nbp, *a1 = df2 + nbp,
           *df1 = a1 + nbp, *gi = df1 + nbp, *wb = gi + nbp, *pp = wb + (size_t)col2 * nbp,
           *sq = pp + (size_t)col2 * nbp, *dM = sq + (size_t)col2 * nbp, *dp0 = dM + (size_t)col2 * col2,
           *ulast = dp0 + col2, *pick = ulast + col2, *dcnt = pick + 4 + 2 * col2,
           *dmap = dcnt + nranks, *L = dmap + 80, *G = L + (size_t)narr_l * lbp;
    // M as a dense matrix: column a = bmv(e_a)   (host, O(col^3))
    std::vector<double> M((size_t)col2 * col2), e(col2), out(col2);
    for (int a = 0; a < col2; ++a) {
      std::fill(e.begin(), e.end(), 0.0);
      e[a] = 1.0;
      info = lbh::bmv(m, sy.data(), wt.data(), col, e.data(), out.data());
      if (info != 0) return 0;
      for (int b = 0; b < col2; ++b) M[(size_t)b + (size_t)a * col2] = out[b];
    }
    HIPCHK(hipMemcpyAsync(dM, M.data(), M.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dp0, p0, col2 * sizeof(double), hipMemcpyHostToDevice, stream));
    std::vector<int> map(narr_l);
    if (multi) {
      // where array a of a rank's gathered block goes among the scan arrays (units of nbp)
      map[0] = 0, map[1] = 1, map[2] = 2, map[3] = 6;
      for (int cc = 0; cc < col2; ++cc) map[4 + cc] = 7 + cc, map[4 + col2 + cc] = 7 + col2 + cc;
      HIPCHK(hipMemcpyAsync(dcnt, counts.data(), nranks * sizeof(double), hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(dmap, map.data(), narr_l * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));  // (M, p0, counts, map are host temporaries)
    if (!multi) {
      lbk::launch_pgcp_gather<T>(q, idx[1], keys[1], nb, nbp, x, l, u, g, Wc(), head, col, theta, r,
                                 d_src(), pend, tt, dd, a0, wb, pp, (double *)nullptr, row0);
    } else {
      // own records in local order -> all ranks -> merged by (t, global index): the merge sort is
      // stable and ranks own ascending row blocks, so equal t keep global index order
      double *Lt = L, *Ld = L + lbp, *La = L + 2 * lbp, *Lg = L + 3 * lbp, *Lw = L + 4 * lbp,
             *Lu = Lw + (size_t)col2 * lbp;
      HIPCHK(hipMemsetAsync(L, 0, (size_t)narr_l * lbp * sizeof(double), stream));
      if (cnt)
        lbk::launch_pgcp_gather<T>(q, idx[1], keys[1], (int64_t)cnt, lbp, x, l, u, g, Wc(), head, col, theta,
                                   r, d_src(), pend, Lt, Ld, La, Lw, Lu, Lg, row0);
      CHK(allgather_big(L, G, (size_t)narr_l * lbp));
      const size_t slots = (size_t)nranks * lbp;
      lbk::launch_pgcp_mergekeys(q, nranks, lbp, narr_l, dcnt, G, keys[0], idx[0]);
      lbk::launch_sort_pairs(q, sort_tmp, sort_tmp_bytes, keys[0], keys[1], idx[0], idx[1], slots);
      lbk::launch_pgcp_permute(q, nb, nbp, lbp, narr_l, idx[1], G, pg_buf, (const int *)dmap);
    }
    lbk::launch_pgcp_last(q, nb, nbp, col2, pp, ulast);
    for (int cc = 0; cc < col2; ++cc)
      lbk::launch_scan(q, pg_tmp, pg_tmp_bytes, pp + (size_t)cc * nbp, pp + (size_t)cc * nbp, (size_t)nb, 1);
    lbk::launch_pgcp_dtp(q, nb, nbp, col2, tt, pp, sq);
    for (int cc = 0; cc < col2; ++cc)
      lbk::launch_scan(q, pg_tmp, pg_tmp_bytes, sq + (size_t)cc * nbp, sq + (size_t)cc * nbp, (size_t)nb, 0);
    lbk::launch_pgcp_terms(q, nb, nbp, col2, theta, dM, dp0, tt, dd, a0, wb, pp, sq, df2, a1);
    // f2 after every breakpoint, with the clamp f2 = max(epsmch f2_org, .) of :1483 (df2 -> F2 in place)
    const double eps_clamp = (sizeof(T) == 4 ? (double)std::numeric_limits<float>::epsilon()
                                             : std::numeric_limits<double>::epsilon()) * f2_org;
    lbk::launch_pgcp_f2(q, pg_tmp, pg_tmp_bytes, nb, f2_0, eps_clamp, df2, dd, df2);
    lbk::launch_pgcp_f1(q, nb, f2_0, tt, df2, a1, df1);
    lbk::launch_scan(q, pg_tmp, pg_tmp_bytes, df1, df1, (size_t)nb, 0);
    lbk::launch_pgcp_find(q, nb, f1_0, f2_0, tt, df1, df2);
    CHK(fetch(0, 1, 0));
    const int64_t ks = h_res[0] < (double)nb ? (int64_t)h_res[0] : nb;  // breakpoints crossed
    lbk::launch_pgcp_pick(q, ks, nb, nbp, col2, f1_0, f2_0, tt, df1, df2, pp, ulast, sq, idx[1],
                          multi ? gi : (const double *)nullptr, pick);
    std::vector<double> pk(4 + 2 * (size_t)col2);
    if (multi) {  // every rank continues from rank 0's numbers, bit for bit
      HIPCHK(hipMemcpyAsync(d_msg, pick, pk.size() * sizeof(double), hipMemcpyDeviceToDevice, stream));
      CHK(exchange(pk.size()));
      std::memcpy(pk.data(), h_msg_all, pk.size() * sizeof(double));
    } else {
      HIPCHK(hipMemcpyAsync(pk.data(), pick, pk.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      nsync++;
    }
    const double t_last = pk[0], f1p = pk[1], f2p = pk[2];
    const int64_t i_last = ks > 0 ? (multi ? 0 : row0) + (int64_t)pk[3] : -1;
    double dtm;
    bool all_fixed = false;
    if (ks < nb) {
      dtm = -f1p / f2p;
    } else if (nb == nglob) {  // every variable fixed (:1436-1442)
      dtm = 0.0;
      all_fixed = true;
    } else if (bnded) {
      dtm = 0.0;
    } else {
      dtm = -f1p / f2p;
    }
    if (debug_walk)
      std::fprintf(stderr, "[pgcp r%d] nb=%lld ks=%lld t_last=%.17g i_last=%lld f1=%.17g f2=%.17g dtm=%.17g p0[0]=%.17g f1_0=%.17g f2_0=%.17g\n",
                   rank, (long long)nb, (long long)ks, t_last, (long long)i_last, f1p, f2p, dtm, p0[0], f1_0, f2_0);
    if (dtm <= 0.0) dtm = 0.0;
    const double tsum = t_last + dtm;
    for (int a = 0; a < col2; ++a)
      c[a] = (t_last * p0[a] - pk[4 + col2 + a]) + dtm * (p0[a] - pk[4 + a]);
    const int64_t ns = 1 + ks - (all_fixed ? 1 : 0);
    nseg = (int)std::min<int64_t>(ns, std::numeric_limits<int>::max());
    // iwhere and z by the cursor: everything up to the last crossed breakpoint is fixed
    gcp = Gcp{};
    gcp.tsum = tsum, gcp.last_t = ks > 0 ? t_last : -1.0, gcp.last_i = i_last;
    lbk::launch_cauchy_finish<T>(q, n, row0, x, l, u, g, tbrk, iwhere, z, tsum, gcp.last_t, gcp.last_i);
    iw_dirty += 1.0;
    z_valid = true;
    done = true;
    return 0;
  }

  // an n-vector on the host, for the iprint >= 100 dumps (debugging sizes, this rank's rows)
  std::vector<double> host_vec(const T *dptr) {
    std::vector<T> tmp((size_t)n);
    (void)hipMemcpyAsync(tmp.data(), dptr, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    return std::vector<double>(tmp.begin(), tmp.end());
  }
