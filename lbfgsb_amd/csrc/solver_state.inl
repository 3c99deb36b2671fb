// solver_state.inl -- member functions of Solver<T> (included inside the class body in solver.hip):
// export / import of the state in the reference's wa / iwa layout (:246-284), the per-kernel doors of
// the C ABI, communicator attachment.
  // ============================================================ state exchange
  int export_state(void *wa_, int32_t *iwa) override {
    // several ranks: every rank exports ITS rows in the same layout (n = n_local); the host matrices
    // are replicated, Index is the local list and -- the global counters of isave not telling how
    // many of THIS rank's rows are free -- Indx2(1) carries the local free count
    if (nranks != 1 && index)
      return fail(LBFGSB_E_STATE, "export_state: contexts that mirror Index are single-rank");
    if (defer_live)
      return fail(LBFGSB_E_STATE, "export_state: the line-search set-up of this 'FG_LNSRCH' return is still "
                                  "deferred (LBFGSB_F_DEFER_LNSRCH): export at a NEW_X return");
    HIPCHK(hipSetDevice(device));
    (void)W();  // (the reference's wa holds the columns in natural row order; the layout is re-made by the policy)
    T *wa = (T *)wa_;
    const int64_t mn = (int64_t)m * n, mm = (int64_t)m * m;
    HIPCHK(hipMemcpy2DAsync(wa, (size_t)n * sizeof(T), ws, (size_t)ld * sizeof(T),
                            (size_t)n * sizeof(T), m, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpy2DAsync(wa + mn, (size_t)n * sizeof(T), wy, (size_t)ld * sizeof(T),
                            (size_t)n * sizeof(T), m, hipMemcpyDeviceToHost, stream));
    T *ps = wa + 2 * mn;
    auto put = [&](const std::vector<double> &v) {
      for (double e : v) *ps++ = (T)e;
    };
    put(sy), put(ss), put(wt), put(wn), put(snd);
    (void)mm;
    // (z and d left implicit by a lean subspace pass: written out for the export only -- the
    //  state of the run does not change, both buffers are dead storage while d_impl stands)
    if (d_impl && x_lean)
      lbk::launch_dz_materialise<T>(q, n, x_lean, t, d, z_in_x ? z : (T *)nullptr);
    for (T *src : {z, r, d, t, xp}) {
      HIPCHK(hipMemcpyAsync(ps, src, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, stream));
      ps += n;
    }
    put(wa8m);
    if (iwa) {
      if (index) {
        HIPCHK(hipMemcpyAsync(iwa, index, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(iwa + 2 * n, indx2, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
      }
    }
    HIPCHK(hipStreamSynchronize(stream));
    if (iwa) {  // iwhere: one byte per row on the device, int32 in the reference's layout
      std::vector<lbk::iw_t> h((size_t)n);
      HIPCHK(hipMemcpy(h.data(), iwhere, (size_t)n * sizeof(lbk::iw_t), hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < n; ++i) iwa[n + i] = h[(size_t)i];
      if (!index) {
        // Contexts that do not mirror the reference's lists keep only the MEMBERSHIP of the free
        // set as of the last freev (wasfree): Index is rebuilt from it in freev's order (:2044-
        // 2054: free variables ascending from the front, active ones from the back).  The
        // enter/leave segments of Indx2 are dead outside the call that made them (formk reads
        // them in the same call, the next freev overwrites them): exported as zeros.
        std::memset(iwa, 0, (size_t)n * 4);
        std::memset(iwa + 2 * n, 0, (size_t)n * 4);
        if (index_valid) {
          std::vector<int8_t> wf((size_t)n);
          HIPCHK(hipMemcpy(wf.data(), wasfree, (size_t)n, hipMemcpyDeviceToHost));
          int64_t nf = 0, ia = n;
          for (int64_t i = 0; i < n; ++i) {
            if (wf[(size_t)i])
              iwa[nf++] = (int32_t)(i + 1);
            else
              iwa[--ia] = (int32_t)(i + 1);
          }
          if (nranks != 1) iwa[2 * n] = (int32_t)nf;
        }
      }
    }
    return 0;
  }

  int import_state(const void *wa_, const int32_t *iwa, const int32_t *isave_user) override {
    if (nranks != 1 && index)
      return fail(LBFGSB_E_STATE, "import_state: contexts that mirror Index are single-rank");
    HIPCHK(hipSetDevice(device));
    // validate BEFORE anything of the context is overwritten: an E_STATE return leaves it as it was
    std::vector<int8_t> wf((size_t)n, 0);
    const int64_t nfree_glob = isave_user[37];
    const int64_t nfree = nranks != 1 ? (int64_t)iwa[2 * n] : nfree_glob;  // (see export_state)
    bool have_index = false;
    for (int64_t i = 0; i < n && !have_index; ++i) have_index = iwa[i] != 0;
    if (!have_index) {  // state from before the first freev (START / FG_START)
      std::fill(wf.begin(), wf.end(), (int8_t)1);
    } else {
      if (nfree < 0 || nfree > n) return fail(LBFGSB_E_STATE, "import_state: isave(38) (nfree) out of range");
      for (int64_t i = 0; i < nfree; ++i) {
        const int64_t k = iwa[i];
        if (k < 1 || k > n) return fail(LBFGSB_E_STATE, "import_state: Index entry out of range");
        wf[(size_t)(k - 1)] = 1;
      }
    }
    const T *wa = (const T *)wa_;
    const int64_t mn = (int64_t)m * n;
    // (the imported columns are in natural row order)
    lbk::launch_lmask_ones(q, n, lmask);
    cw_packed = false, cw_stale = 0, cw_hold = 0, cw_settled = 0;
    live_head = std::min(std::max(isave_user[26], 1), m), live_col = std::min(std::max(isave_user[27], 0), m);
    HIPCHK(hipMemcpy2DAsync(ws, (size_t)ld * sizeof(T), wa, (size_t)n * sizeof(T),
                            (size_t)n * sizeof(T), m, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpy2DAsync(wy, (size_t)ld * sizeof(T), wa + mn, (size_t)n * sizeof(T),
                            (size_t)n * sizeof(T), m, hipMemcpyHostToDevice, stream));
    const T *ps = wa + 2 * mn;
    auto get = [&](std::vector<double> &v) {
      for (double &e : v) e = (double)*ps++;
    };
    get(sy), get(ss), get(wt), get(wn), get(snd);
    t = t_own, r = r_own;  // (the imported t and r live in the context's own buffers, whichever entry is used)
    entry_mode = 0;        // (either device-pointer entry may continue the imported run)
    ub_mask = 0;           // (uniform bounds are detected by a START only)
    iw_dirty = 1.0;        // (the imported iwhere has not been through a freev of this context)
    for (T *dst : {z, r, d, t, xp}) {
      HIPCHK(hipMemcpyAsync(dst, ps, (size_t)n * sizeof(T), hipMemcpyHostToDevice, stream));
      ps += n;
    }
    get(wa8m);
    z_valid = true;  // z as imported
    spec.valid = false, pend.on = 0, pend.impl = 0, d_impl = z_in_x = false, tbrk_valid = false, scan.ready = false;
    ls.deferred = false, defer_live = false;
    sfv.valid = false, sfv_hot = false, eager.valid = false, spec_live_len = 0;
    nbd8_src = nullptr;
    spcand.valid = false;
    {
      std::vector<lbk::iw_t> h((size_t)n);
      for (int64_t i = 0; i < n; ++i) h[(size_t)i] = (lbk::iw_t)iwa[n + i];
      HIPCHK(hipMemcpy(iwhere, h.data(), (size_t)n * sizeof(lbk::iw_t), hipMemcpyHostToDevice));
    }
    // free-set membership as of the last freev: Index(1:nfree), validated above
    index_valid = have_index;
    HIPCHK(hipMemcpyAsync(wasfree, wf.data(), (size_t)n, hipMemcpyHostToDevice, stream));
    if (index) {
      HIPCHK(hipMemcpyAsync(index, iwa, (size_t)n * 4, hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(indx2, iwa + 2 * n, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));
    nfree_g = nfree_glob;
    nenter_g = isave_user[40];
    ileave_g = isave_user[39];
    return 0;
  }

  // ======================================================= per-kernel entries
  int k_projgr(const void *x, const void *l, const void *u, const int32_t *nbd, const void *g,
               double *out) override {
    HIPCHK(hipSetDevice(device));
    lbk::launch_projgr<T>(q, n, (const T *)x, (const T *)l, (const T *)u, nbd, (const T *)g);
    CHK(fetch(0, 0, 1));
    *out = h_res[0];
    return 0;
  }
  int k_wtv(const void *v, int col, int head, double *out, bool launch_only) override {
    HIPCHK(hipSetDevice(device));
    if (col < 1 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "wtv: bad col/head");
    if (col > lbk::MAXM) {  // beyond the fused width: tile by tile
      if (launch_only) return fail(LBFGSB_E_ARG, "wtv: launch-only timing is for col <= 32");
      return wide_wtv((const T *)v, col, head, out, out + col);
    }
    if (launch_only) {
      lbk::launch_wtv_nofinalize<T>(q, n, W(), head, col, (const T *)v);
      return 0;
    }
    lbk::launch_wtv<T>(q, n, W(), head, col, (const T *)v);
    const int MC = lbk::maxc_for(col);
    CHK(fetch(2 * MC, 0, 0));
    for (int j = 0; j < col; ++j) {
      out[j] = h_res[j];
      out[col + j] = h_res[MC + j];
    }
    return 0;
  }
  int k_launch(int which, const void *x, const void *g, int col, int head) override {
    if (col < 1 || col > m || head < 1 || head > m || col > lbk::MAXM) return fail(LBFGSB_E_ARG, "bad col/head");
    HIPCHK(hipSetDevice(device));
    lbk::Coef cf;
    std::memset(&cf, 0, sizeof cf);
    if (which == 0 || which == 2)
      lbk::launch_cmprlb_wtv<T>(q, n, (const T *)x, (const T *)g, 0.5, iwhere, W(), head, col, 1.0,
                                cf, which == 2 ? 1 : 0, r_own, d, lbk::Pend{1, 0.5, 0});
    else if (which == 1)
      lbk::launch_formk_gram<T>(q, n, W(), head, col, iwhere);
    else if (which == 3 || which == 4) {
      if (!cl || !cu || !cnbd) return fail(LBFGSB_E_STATE, "kernel_time: run an iteration first");
      const T *l = (const T *)cl, *u = (const T *)cu;
      // (the variants the iteration launches; the lean subspace pass stores its trial point into
      //  the z buffer here instead of the caller's x -- the same five store streams)
      const bool lean = lean_on && !(flags & LBFGSB_F_MIRROR_INDEX);
      if (which == 3)  // with a pending pair: the variant every iteration after an update runs
        lbk::launch_subsm_update<T>(q, n, 0.5, lean ? (T *)nullptr : z, r_own, pp ? (T *)nullptr : r_own, lk(l),
                                    uk(u), nbk(), iwhere, (const T *)x, (const T *)g, Wc(), head, col, 1.0, cf,
                                    cf, lean ? (T *)nullptr : d, pp ? (T *)nullptr : t_own,
                                    lean ? z : (T *)nullptr, 1, lbk::Pend{1, 0.5, lean ? 1 : 0},
                                    lean ? t_own : d, ub_mask);
      else {           // as the evaluation of a trial point: reduces only
        // (on the packed layout the direction is taken as x - x = 0: a stale t_own would make EVERY row look as if
        //  it had moved, and rows outside their tile's front run would all fetch their entries the slow way -- the
        //  steady state this door is meant to time has a handful of such rows)
        const bool pk = Wc().lmask != nullptr;
        lbk::launch_update_scan<T>(q, n, (const T *)x, lk(l), uk(u), nbk(), (const T *)g, r_own,
                                   pk ? (const T *)x : (lean ? t_own : d), (pk || lean) ? 1 : 0, 0.5, iwhere,
                                   (T *)nullptr, Wc(), head, col,
                                   (head + col - 2) % m + 1, 0, 0, nr_flag(col), -1.0, nullptr, nullptr, 0,
                                   nullptr, ub_mask);
      }
    } else
      return fail(LBFGSB_E_ARG, "unknown kernel");
    return 0;
  }
  int k_set_w(const void *hws, const void *hwy) override {
    HIPCHK(hipSetDevice(device));
    lbk::launch_lmask_ones(q, n, lmask);  // (natural row order)
    cw_packed = false, cw_stale = 0, live_head = 1, live_col = m;
    HIPCHK(hipMemcpy2DAsync(ws, (size_t)ld * sizeof(T), hws, (size_t)n * sizeof(T),
                            (size_t)n * sizeof(T), m, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpy2DAsync(wy, (size_t)ld * sizeof(T), hwy, (size_t)n * sizeof(T),
                            (size_t)n * sizeof(T), m, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  int k_set_iwhere(const int32_t *h_iw) override {
    HIPCHK(hipSetDevice(device));
    std::vector<lbk::iw_t> h((size_t)n);
    for (int64_t i = 0; i < n; ++i) h[(size_t)i] = (lbk::iw_t)h_iw[i];
    HIPCHK(hipMemcpy(iwhere, h.data(), (size_t)n * sizeof(lbk::iw_t), hipMemcpyHostToDevice));
    iw_dirty = 1.0;
    return 0;
  }
  int k_formk_gram(int col, int head, double *out) override {
    HIPCHK(hipSetDevice(device));
    if (col < 1 || col > m || head < 1 || head > m || col > lbk::MAXM)
      return fail(LBFGSB_E_ARG, "formk_gram: bad col/head");
    lbk::launch_formk_gram<T>(q, n, W(), head, col, iwhere);
    const int E = 2 * col * col + col;
    CHK(fetch(E, 0, 0));
    std::memcpy(out, h_res, sizeof(double) * E);
    return 0;
  }
  int k_objective(int kind, const void *x, void *g, double *f) override {
    HIPCHK(hipSetDevice(device));
    // (f == nullptr: the value is fetched with the next call's first pass -- its partials go to a matrix of
    //  their own and that pass's finalize takes them along)
    if (!f && fold_fin) q.part_sel = 1, q.hold_fin = true;
    struct Reset {
      lbk::Queue &q;
      ~Reset() { q.part_sel = 0, q.hold_fin = false; }
    } reset{q};
    if (kind == 0) {
      lbk::launch_obj_quadratic<T>(q, n, row0, (const T *)x, (T *)g);
    } else if (kind == 1) {
      if (nglob < 2) return fail(LBFGSB_E_ARG, "rosenbrock objective needs n >= 2");
      double xl = 0.0, xr = 0.0;
      if (nranks > 1) {  // 1-element halo: every rank's first and last x, all-gathered
        lbk::launch_halo_pack<T>(q, n, (const T *)x, d_msg);
        CHK(exchange(2));
        if (rank > 0) xl = h_msg_all[2 * (rank - 1) + 1];
        if (rank < nranks - 1) xr = h_msg_all[2 * (rank + 1)];
      }
      lbk::launch_obj_rosenbrock<T>(q, n, row0, nglob, (const T *)x, (T *)g, xl, xr);
    } else {
      return fail(LBFGSB_E_ARG, "unknown objective kind");
    }
    f_scale = kind == 0 ? 0.5 : 4.0;
    if (!f) {  // deferred: no host sync here
      f_pending = true;
      return 0;
    }
    CHK(fetch(1, 0, 0));
    *f = f_scale * h_res[0];
    return 0;
  }
  // the caller's own objective value, left on the device (this rank's part of f; fp64): it rides to the host with
  // the next setulb call's first fetch exactly as a built-in objective's value does -- no host sync for f
  int f_device(const double *d_f) override {
    HIPCHK(hipSetDevice(device));
    if (fold_fin) q.part_sel = 1, q.hold_fin = true;
    lbk::launch_scalar_partial(q, d_f);
    q.part_sel = 0, q.hold_fin = false;
    f_scale = 1.0;
    f_pending = true;
    return 0;
  }
  int sync() override {
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  int attach_rccl(ncclComm_t c, int rank_, int nranks_) override {
    if (comm) g_rccl.CommDestroy(comm);  // (a second init replaces the communicator)
    comm = nullptr;
    CHK(set_ranks(rank_, nranks_));
    comm = c;
    q.fin_publish = false;  // (the results travel through the all-gather; publish_kernel follows it)
    return 0;
  }
  ncclComm_t rccl_comm() const override { return comm; }
  int attach_host(lbfgsb_allreduce_fn ar, lbfgsb_allgather_fn ag, void *user, int rank_,
                  int nranks_) override {
    cb_ar = ar, cb_ag = ag, cb_user = user;
    return set_ranks(rank_, nranks_);
  }
  void path_counts(int64_t &closed_form, int64_t &three_pass) const override {
    closed_form = nclosed, three_pass = nthreepass;
  }
  int64_t freev_skipped() const override { return nfreev_skipped; }
  void compact_stats(int64_t &packs, int64_t &unpacks, int &packed, int &eligible) const override {
    packs = ncw_pack, unpacks = ncw_unpack, packed = cw_packed ? 1 : 0, eligible = cw_eligible() ? 1 : 0;
  }
  // one host sync of the iteration, timed by itself: the 8 min(m, 32) + 15 partials of the widest phase through
  // fetch() -- with a communicator the all-gather over the ranks on the solver's stream, the copy into mapped host
  // memory and the poll for it; without one the publish + poll alone.  Every rank must call this (it IS a
  // collective).  Not inside a run (between the returns of one: nothing may be deferred / pending).
  int collective_time(int reps, double *median_us, double *min_us) override {
    if (reps < 1 || reps > 100000) return fail(LBFGSB_E_ARG, "collective_time: reps out of range");
    if (defer_live || spec_live_len) return fail(LBFGSB_E_STATE, "collective_time: sums of a run are in flight");
    HIPCHK(hipSetDevice(device));
    const int k = std::min<int>(8 * std::min(m, lbk::MAXM) + 15, lbk::RES_MAX);
    const int64_t ns0 = nsync, nc0 = ncoll, cb0 = coll_bytes;
    const double tw0 = t_wait;
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<double> t((size_t)reps);
    for (int it = 0; it < reps; ++it) {
      const double t0 = now_s();
      CHK(fetch(k, 0, 0));
      t[(size_t)it] = (now_s() - t0) * 1e6;
    }
    nsync = ns0, ncoll = nc0, coll_bytes = cb0, t_wait = tw0;  // (a measurement, not part of any run's counts)
    std::sort(t.begin(), t.end());
    if (median_us) *median_us = t[t.size() / 2];
    if (min_us) *min_us = t.front();
    return 0;
  }
  int64_t skip_scans_reused() const override { return nskip_reused; }
  void defer_counts(int64_t &deferred, int64_t &reissued) const override { deferred = ndeferred, reissued = nredo; }
  const void *prev_iterate() const override { return t; }
  int uniform_mask() const override { return ub_mask; }
  int bounds_same(const void *l0, const void *u0, const int32_t *nb0, const void *l1, const void *u1,
                  const int32_t *nb1, double *ndiff) override {
    HIPCHK(hipSetDevice(device));
    lbk::launch_bounds_same<T>(q, n, (const T *)l0, (const T *)u0, nb0, (const T *)l1, (const T *)u1, nb1);
    CHK(fetch(1, 0, 0));
    *ndiff = h_res[0];
    return 0;
  }
