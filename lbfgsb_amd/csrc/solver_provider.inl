// solver_provider.inl -- member functions of Solver<T> (included inside the class body in solver.hip):
// the breakpoint PROVIDER of the generalized Cauchy point (reference src/lbfgsb.f90:1157-1532) -- window
// compaction, (t, index) ordering, chunked record gathers, the all-gather + merge over ranks, the
// reference's own heap order for walks that end inside a group of equal breakpoints -- and the small
// pieces of state the Cauchy phase shares with the other phases (pending pair, lean d / z, nbd8).
// The exact host replay of the walk (:1378-1497) is in solver_walk.inl, the opt-in parallel search
// (LBFGSB_F_PARALLEL_GCP) in solver_pgcp.inl.
  // =================================================================== cauchy
  // Breakpoint provider: hands the replicated host walk the breakpoints of ALL ranks in
  // ascending (t, global index) order (SURVEY.md 7.3-1 option (a)).  Each rank keeps its own
  // candidates sorted on the device; chunks of records are all-gathered and merged on the
  // host.  A merged record is "safe" to consume once no rank can still hold an earlier one.
  struct MRec {
    double t;
    int64_t gidx;
    int rank;
    const double *rec;
  };
  struct Provider {
    bool have = false;   // candidate lists exist on the devices
    bool full = false;   // lists = ALL remaining breakpoints (full sort)
    double win_hi = -1;  // lists cover every breakpoint after the fetch cursor with t <= win_hi
    uint32_t Cl = 0;     // local list length
    uint32_t pl = 0;     // local list position of the first record not yet consumed
    int cur = 0;         // which keys/idx buffer holds the sorted local list
    std::vector<MRec> M; // merged chunk, all ranks
    const double *raw = nullptr;  // col = 0 and the chunk itself is in order (single rank, or merged
                                  // on the device): records of 4 doubles; M is then not touched at
                                  // all (raw_n records)
    size_t raw_n = 0;
    const unsigned char *rank_of = nullptr;  // device-merged chunk: the rank every record came from
    size_t msize() const { return raw ? raw_n : M.size(); }
    size_t mpos = 0, safe_end = 0;
    bool more_anywhere = false;
    std::vector<uint32_t> taken;
    uint32_t next_chunk = 64;
    int grow = 0;
    // the reference's own pop order (bkmin first, then hpsolb's heap), replayed on the host over
    // ALL breakpoints; records are gathered in that order
    bool exact = false;
    std::vector<double> ht;       // heap keys   (t of hpsolb, 0-based)
    std::vector<uint32_t> hio;    // heap values (iorder: GLOBAL rows), n_global < 2^32 ...
    std::vector<int64_t> hio64;   // ... and beyond (h64)
    bool h64 = false;
    std::vector<int64_t> hrow0;   // first global row of every rank (+ nglob at the end)
    int64_t hleft = 0;            // nleft of the reference's walk for the NEXT pop
    bool hbuilt = false;
    int64_t hibkmin = -1;
  };

  // every rank contributes d_msg[0..count) (device); all of it lands in h_msg_all (rank-major)
  int exchange(size_t count) {
    if (comm || nranks > 1) ncoll++, coll_bytes += (int64_t)count * 8;
    if (nranks == 1 && !comm) {
      HIPCHK(hipMemcpyAsync(h_msg_all, d_msg, count * sizeof(double), hipMemcpyDeviceToHost,
                            stream));
      {
        const double t0 = now_s();
        HIPCHK(hipStreamSynchronize(stream));
        t_wait += now_s() - t0;
      }
    } else if (comm) {
      if (g_rccl.AllGather(d_msg, d_msg_all, count, ncclDouble, comm, stream) != ncclSuccess)
        return fail(LBFGSB_E_COMM, "ncclAllGather failed");
      HIPCHK(hipMemcpyAsync(h_msg_all, d_msg_all, (size_t)nranks * count * sizeof(double),
                            hipMemcpyDeviceToHost, stream));
      {
        const double t0 = now_s();
        HIPCHK(hipStreamSynchronize(stream));
        t_wait += now_s() - t0;
      }
    } else {
      if (!cb_ag) return fail(LBFGSB_E_COMM, "multi-rank context without an all-gather");
      HIPCHK(hipMemcpyAsync(h_msg_loc, d_msg, count * sizeof(double), hipMemcpyDeviceToHost,
                            stream));
      {
        const double t0 = now_s();
        HIPCHK(hipStreamSynchronize(stream));
        t_wait += now_s() - t0;
      }
      if (cb_ag(cb_user, h_msg_loc, h_msg_all, (int64_t)(count * sizeof(double))) != 0)
        return fail(LBFGSB_E_COMM, "host all-gather callback failed");
    }
    nsync++;
    return 0;
  }
  int put_header(double a, double b) {
    h_hdr[0] = a, h_hdr[1] = b;
    HIPCHK(hipMemcpyAsync(d_msg, h_hdr, 2 * sizeof(double), hipMemcpyHostToDevice, stream));
    return 0;
  }
  // breakpoint times as a vector: written by cauchy_scan_kernel; the fused update pass does
  // not store them (the usual short walk recomputes the few it needs), so the rare consumers
  // of the vector (full sort, cursor-based cauchy_finish) fill it in first
  bool tbrk_valid = false;
  const int32_t *cnbd = nullptr;
  // nbd as one byte per row for the passes over W (lbk::nb_t): packed when a run starts, when a
  // state is imported, and whenever the caller's pointer changes.  Like l and u, nbd must not
  // change between START and the end of a run (the reference reads it afresh on every call,
  // but a run whose bound types change under it has no meaning there either).
  lbk::nb_t *nbd8 = nullptr;
  const int32_t *nbd8_src = nullptr;
  int ensure_nbd8(const int32_t *nbd) {
    if (nbd8_src == nbd) return 0;
    lbk::launch_nbd_pack(q, n, nbd, nbd8);
    nbd8_src = nbd;
    return 0;
  }
  // the pair accepted by matupd in this call, not yet stored in W (see lbk::Pend)
  lbk::Pend pend{0, 1.0, 0};
  // ---- lean subspace pass: z and d stay implicit (z = x, d = x - t) while the unit first trial
  //      step stands; ensure_d() writes them out for everything but the hot path ----
  bool d_impl = false;
  bool z_in_x = false;  // ... and z too: until the next cauchy gives z a new meaning
  bool lean_on = true;  // (option "lean")
  const T *d_src() const { return d_impl ? t : d; }  // what the kernels read the direction from
  // (xnew: the line search's next trial point x = stp d + t written by the same pass, see dz_materialise_kernel;
  //  returns with *stepped = true if it was)
  int ensure_d(const T *x, T *xnew = nullptr, double stp_new = 1.0, bool *stepped = nullptr) {
    if (stepped) *stepped = false;
    if (!d_impl) return 0;
    if (xnew && stp_new == 1.0) xnew = nullptr;  // (a unit step copies z bit for bit: lnsrlb_step_kernel)
    lbk::launch_dz_materialise<T>(q, n, x, t, d, z_in_x ? z : (T *)nullptr, xnew, stp_new);
    if (stepped) *stepped = xnew != nullptr;
    if (z_in_x) z_valid = true;
    d_impl = false, z_in_x = false;
    if (pend.on) pend.impl = 0;
    return 0;
  }
  // sums of an update_scan pass that ran as the evaluation of an accepted trial point (kept
  // from the FG_LNSRCH entry that returned NEW_X to the NEW_X entry that performs the update)
  struct Spec {
    bool valid = false;
    const void *x = nullptr, *g = nullptr;
    double stp = 0.0;
    int head = 0, col = 0, itail = 0;
    std::vector<double> res = std::vector<double>(8 * (size_t)LBFGSB_MAX_M + 64, 0.0);  // (the widest merged layout)
  } spec;
  int commit_pending(const T *g, int col, int head) {
    if (pend.on) {
      CHK(ensure_d((const T *)cx));
      lbk::launch_pair_commit<T>(q, n, g, r, d, pend, W(), head, col);
    }
    pend.on = 0;
    return 0;
  }
  int ensure_tbrk() {
    if (!tbrk_valid)
      lbk::launch_tbrk_fill<T>(q, n, (const T *)cx, (const T *)cl, (const T *)cu, cnbd, (const T *)cg,
                               iwhere, tbrk);
    tbrk_valid = true;
    return 0;
  }
  int local_count(double lo_t, int64_t lo_i, double hi, uint32_t cap, uint32_t &cnt) {
    CHK(ensure_tbrk());
    lbk::launch_cauchy_window<T>(q, n, row0, tbrk, lo_t, lo_i, hi, keys[0], idx[0], cap, d_count);
    HIPCHK(hipMemcpyAsync(h_count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    nsync++;
    cnt = *h_count;
    return 0;
  }

  static constexpr uint32_t FAST_CAP = 256;  // candidates delivered by the one-sync fast path
  // ---- candidates handed over by the update pass itself (update_scan_kernel, cand_hi) ----
  static constexpr uint32_t SPEC_CAP = 128;
  uint64_t *sp_keys = nullptr;
  uint32_t *sp_idx = nullptr, *sp_count = nullptr;
  double *sp_msg = nullptr, *sp_msg_all = nullptr, *h_sp_all = nullptr, *h_sp_loc = nullptr;
  struct SpecCand {
    bool valid = false, fresh = false;
    double hi = -1.0;
    int col = 0;
  } spcand;
  double last_tsum = 0.0, last_dtm0 = 0.0;  // where the previous walk ended / first aimed
  size_t sp_len() const { return 2 + (size_t)SPEC_CAP * (2 * m + 4); }
  double spec_factor = 2.0;
  // The first window of a walk reaches (1 + win_slack) x as far beyond the cursor as the walk needs right now: while
  // it crosses breakpoints its stationary point moves (tj0 + dtm grows by a few per cent as f'' shrinks), and a
  // window that ends exactly at the first estimate is followed by a second pass over x, g for the last one or two
  // breakpoints -- or for none (n = 1e8: 0.3 ms + a host sync each, profiles/round5_D_kernel_trace_*).
  double win_slack = 0.25;
  int64_t last_walk_nseg = 0;  // segments of the previous walk
  // Off unless option "spec_capture" = 1: measured at n = 1e8 / 1.25e7 (profiles/README.md, r02q) a walk
  // either crosses no breakpoint at all or hundreds to thousands -- SPEC_CAP records serve 0-3 of 31.
  bool spec_on = false;
  double spec_hi(bool cnstnd) const {  // the guess: a little beyond where the previous walk ended
    if (!spec_on || !cnstnd || iter_seen < 3) return -1.0;  // (the first walks cross most breakpoints)
    return last_tsum > 0.0 && std::isfinite(last_tsum) ? spec_factor * last_tsum : -1.0;
  }
  int iter_seen = 0;
  // queue the gather of the candidates' records and their way to the host (all ranks') behind the
  // update pass; spec_land() completes it after the phase's one host sync
  int spec_queue(const T *x, const T *l, const T *u, const T *g, int head, int col, double stp) {
    lbk::launch_cauchy_gather_dyn<T>(q, sp_idx, sp_keys, sp_count, SPEC_CAP, row0, x, l, u, g, Wc(), head,
                                     col, r, d_src(), lbk::Pend{1, stp, d_impl ? 1 : 0}, sp_msg);
    const size_t cnt = 2 + (size_t)SPEC_CAP * (2 * col + 4);
    tail_copy_queued = true;  // (the fetch that follows must wait for THIS copy, not for the finalize in front of it)
    if (nranks == 1 && !comm) {
      HIPCHK(hipMemcpyAsync(h_sp_all, sp_msg, cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
    } else if (comm) {
      if (g_rccl.AllGather(sp_msg, sp_msg_all, cnt, ncclDouble, comm, stream) != ncclSuccess)
        return fail(LBFGSB_E_COMM, "ncclAllGather failed");
      HIPCHK(hipMemcpyAsync(h_sp_all, sp_msg_all, (size_t)nranks * cnt * sizeof(double),
                            hipMemcpyDeviceToHost, stream));
    } else {
      HIPCHK(hipMemcpyAsync(h_sp_loc, sp_msg, cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
    }
    return 0;
  }
  size_t sp_stride = 0;  // doubles per rank in h_sp_all
  int spec_land(int col, double hi) {
    const int recl = 2 * col + 4;
    sp_stride = 2 + (size_t)SPEC_CAP * recl;
    if (nranks > 1 && !comm) {
      // host all-gather: first the counts, then only as many records as the fullest rank has
      if (!cb_ag) return fail(LBFGSB_E_COMM, "multi-rank context without an all-gather");
      double *cnts = h_sp_all + (size_t)nranks * sp_len() - nranks;  // (tail of the buffer)
      if (cb_ag(cb_user, h_sp_loc, cnts, (int64_t)sizeof(double)) != 0)
        return fail(LBFGSB_E_COMM, "host all-gather callback failed");
      double mx = 0.0;
      for (int rk = 0; rk < nranks; ++rk) mx = std::max(mx, cnts[rk]);
      const size_t keep = (size_t)std::min<double>(mx, (double)SPEC_CAP);
      sp_stride = 2 + keep * recl;
      if (cb_ag(cb_user, h_sp_loc, h_sp_all, (int64_t)(sp_stride * sizeof(double))) != 0)
        return fail(LBFGSB_E_COMM, "host all-gather callback failed");
    }
    spcand.valid = true, spcand.fresh = true, spcand.hi = hi, spcand.col = col;
    return 0;
  }

  // *big != nullptr: if more than PG_MIN candidates lie in the window, only report their number
  // (the caller switches to the parallel search) instead of ordering them
  // (option "pg_min" lowers it so that tests can send small problems through the search)
  double PG_MIN = 32768.0;
  int window_fetch(Provider &pv, double lo_t, int64_t lo_i, double hi, const T *x, const T *l,
                   const T *u, const T *g, int head, int col, double *big = nullptr) {
    // window compaction + record gather + ONE all-gather/sync: enough for the usual short walk
    const int recl = 2 * col + 4;
    pf_valid = false;  // (new candidate lists: a prefetched chunk of the old ones is void)
    if (debug_walk && lo_t < 0.0) {
      double c0 = spcand.valid ? h_sp_all[0] : -1.0;
      std::fprintf(stderr, "[spec] valid=%d fresh=%d hi_asked=%g spec_hi=%g factor=%g count0=%g\n",
                   (int)spcand.valid, (int)spcand.fresh, hi, spcand.hi, spec_factor, c0);
    }
    if (spcand.valid && spcand.fresh && lo_t < 0.0 && hi > spcand.hi)
      spec_factor = std::min(4.0, spec_factor * 1.5);  // the guess was short: aim further next time
    if (spcand.valid && spcand.fresh && lo_t < 0.0 && hi <= spcand.hi && spcand.col == col) {
      // the update pass already delivered every breakpoint up to spcand.hi with its record
      spcand.fresh = false;
      const size_t scount = sp_stride;
      double gsum = 0.0;
      bool all_in = true;
      for (int rk = 0; rk < nranks; ++rk) {
        const double c = h_sp_all[(size_t)rk * scount];
        gsum += c;
        if (c > (double)SPEC_CAP) all_in = false;
      }
      // adapt the guess: too many candidates -> aim closer next time, few -> a little wider
      if (!all_in)
        spec_factor = std::max(1.05, 0.5 * (spec_factor + 1.0));
      else if (gsum < 0.25 * SPEC_CAP)
        spec_factor = std::min(4.0, spec_factor * 1.25);
      if (all_in) {
        if (big) *big = gsum;
        pv.have = true, pv.full = false;
        pv.win_hi = spcand.hi;
        pv.Cl = (uint32_t)h_sp_all[(size_t)rank * scount];
        pv.pl = pv.Cl;  // everything is already on the host
        pv.cur = 0;
        pv.M.clear();
        pv.raw = nullptr;
        for (int rk = 0; rk < nranks; ++rk) {
          const double *base = h_sp_all + (size_t)rk * scount;
          const uint32_t lr = (uint32_t)base[0];
          for (uint32_t k = 0; k < lr; ++k) {
            const double *rec = base + 2 + (size_t)k * recl;
            pv.M.push_back(MRec{rec[0], (int64_t)rec[1], rk, rec});
          }
        }
        std::sort(pv.M.begin(), pv.M.end(), [](const MRec &a, const MRec &b) {
          return a.t < b.t || (a.t == b.t && a.gidx < b.gidx);
        });
        pv.mpos = 0;
        pv.safe_end = pv.M.size();
        pv.more_anywhere = false;
        pv.taken.assign(nranks, 0);
        pv.next_chunk = 64;
        nspecwin++;
        return 0;
      }
    }
    if (tbrk_valid)
      lbk::launch_cauchy_window<T>(q, n, row0, tbrk, lo_t, lo_i, hi, keys[0], idx[0], SEL_CAP,
                                   d_count);
    else
      lbk::launch_cauchy_window_fly<T>(q, n, row0, x, lk(l), uk(u), nbk(), g, iwhere, lo_t, lo_i, hi, keys[0],
                                       idx[0], SEL_CAP, d_count, ub_mask);
    lbk::launch_cauchy_gather_dyn<T>(q, idx[0], keys[0], d_count, FAST_CAP, row0, x, l, u, g, Wc(),
                                     head, col, r, d_src(), pend, d_msg);
    const size_t fcount = 2 + (size_t)FAST_CAP * recl;
    CHK(exchange(fcount));
    double gsum = 0.0;
    bool all_small = true;
    for (int rk = 0; rk < nranks; ++rk) {
      const double c = h_msg_all[(size_t)rk * fcount];
      gsum += c;
      if (c > (double)FAST_CAP) all_small = false;
    }
    uint32_t cnt = (uint32_t)h_msg_all[(size_t)rank * fcount];
    if (big) {
      *big = gsum;
      if (gsum > PG_MIN) return 0;
    }
    if (all_small) {
      pv.have = true;
      pv.full = false;
      pv.win_hi = hi;
      pv.Cl = cnt;
      pv.pl = cnt;  // everything is already on the host
      pv.cur = 0;
      pv.M.clear();
      pv.raw = nullptr;
      for (int rk = 0; rk < nranks; ++rk) {
        const double *base = h_msg_all + (size_t)rk * fcount;
        const uint32_t lr = (uint32_t)base[0];
        for (uint32_t k = 0; k < lr; ++k) {
          const double *rec = base + 2 + (size_t)k * recl;
          pv.M.push_back(MRec{rec[0], (int64_t)rec[1], rk, rec});
        }
      }
      std::sort(pv.M.begin(), pv.M.end(), [](const MRec &a, const MRec &b) {
        return a.t < b.t || (a.t == b.t && a.gidx < b.gidx);
      });
      pv.mpos = 0;
      pv.safe_end = pv.M.size();
      pv.more_anywhere = false;
      pv.taken.assign(nranks, 0);
      pv.next_chunk = 64;
      return 0;
    }
    pv.have = true;
    pv.pl = 0;
    pv.taken.clear();
    pv.M.clear();
    pv.mpos = pv.safe_end = 0;
    pv.more_anywhere = true;  // forces a refill
    pv.next_chunk = 64;
    if (gsum <= (double)SEL_CAP) {
      pv.full = false;
      pv.win_hi = hi;
      pv.Cl = cnt;
      pv.cur = 0;
      if (cnt > 1) {
        // (t, idx) lexicographic order: stable sort by idx, then stable sort by t
        lbk::launch_sort_by_idx(q, sort_tmp, sort_tmp_bytes, idx[0], idx[1], keys[0], keys[1], cnt);
        lbk::launch_sort_pairs(q, sort_tmp, sort_tmp_bytes, keys[1], keys[0], idx[1], idx[0], cnt);
      }
    } else {
      // too many candidates in the window: order ALL remaining breakpoints once
      nfullsort++;
      CHK(ensure_sel((size_t)n));
      CHK(local_count(lo_t, lo_i, std::numeric_limits<double>::max(), 0, cnt));  // (fills tbrk)
      lbk::launch_cauchy_allkeys<T>(q, n, row0, tbrk, lo_t, lo_i, keys[0], idx[0]);
      lbk::launch_sort_pairs(q, sort_tmp, sort_tmp_bytes, keys[0], keys[1], idx[0], idx[1],
                             (size_t)n);
      pv.full = true;
      pv.win_hi = std::numeric_limits<double>::infinity();
      pv.Cl = cnt;  // the rest of the sorted array are non-candidates (key = ~0)
      pv.cur = 1;
    }
    return 0;
  }

  // ---- breakpoints in the reference's own order ----
  // cauchy takes the smallest breakpoint from the scan (first minimum in variable order, :1384-
  // 1389), then moves the last list entry into its slot, builds hpsolb's heap over the rest and
  // pops one breakpoint per segment (:1391-1403).  Among EQUAL breakpoints that order is a
  // property of the heap, not of the variables; it matters only when the walk ends inside a
  // group of equal breakpoints (then it decides which of them are fixed).  Replaying it needs the
  // whole list on the host: O(n) transfer + heap build, so it runs only for a call whose walk did
  // end inside such a group (or from the start under iprint >= 99); LBFGSB_F_INDEX_TIES opts out.
  int exact_init(Provider &pv) {
    CHK(ensure_tbrk());
    std::vector<T> tb((size_t)n);
    HIPCHK(hipMemcpyAsync(tb.data(), tbrk, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    nsync++;
    pv = Provider{};
    pv.exact = true;
    pv.h64 = nglob >= 0xffffffffll;
    pv.ht.clear(), pv.hio.clear(), pv.hio64.clear();
    const double inf = std::numeric_limits<double>::infinity();
    // every rank's breakpoint times, in global variable order (ranks own ascending row blocks)
    std::vector<double> tall;
    std::vector<int64_t> cnt(nranks, n);
    pv.hrow0.assign((size_t)nranks + 1, 0);
    int64_t nmax = n;
    if (nranks > 1) {
      CHK(put_header((double)n, (double)row0));
      CHK(exchange(2));
      nmax = 0;
      for (int rk = 0; rk < nranks; ++rk) {
        cnt[rk] = (int64_t)h_msg_all[2 * (size_t)rk];
        pv.hrow0[rk] = (int64_t)h_msg_all[2 * (size_t)rk + 1];
        nmax = std::max(nmax, cnt[rk]);
      }
      std::vector<double> mine((size_t)nmax, -1.0);
      for (int64_t i = 0; i < n; ++i) mine[(size_t)i] = (double)tb[(size_t)i];
      double *dsend = nullptr, *drecv = nullptr;
      HIPCHK(hipMalloc(&dsend, (size_t)nmax * sizeof(double)));
      if (hipMalloc(&drecv, (size_t)nmax * nranks * sizeof(double)) != hipSuccess) {
        (void)hipFree(dsend);
        return fail(LBFGSB_E_NOGPU, "exact tie order: no memory for the gathered breakpoint times");
      }
      tall.resize((size_t)nmax * nranks);
      int rc = 0;
      if (hipMemcpy(dsend, mine.data(), (size_t)nmax * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(LBFGSB_E_NOGPU, "exact tie order: upload failed");
      if (!rc) rc = allgather_big(dsend, drecv, (size_t)nmax);
      if (!rc && hipStreamSynchronize(stream) != hipSuccess) rc = fail(LBFGSB_E_NOGPU, "exact tie order: sync");
      if (!rc && hipMemcpy(tall.data(), drecv, tall.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(LBFGSB_E_NOGPU, "exact tie order: download failed");
      (void)hipFree(dsend), (void)hipFree(drecv);
      if (rc) return rc;
    } else {
      pv.hrow0[0] = row0;
      tall.resize((size_t)n);
      for (int64_t i = 0; i < n; ++i) tall[(size_t)i] = (double)tb[(size_t)i];
    }
    pv.hrow0[nranks] = nglob;
    double bk = 0.0;
    for (int rk = 0; rk < nranks; ++rk)
      for (int64_t i = 0; i < cnt[rk]; ++i) {  // the list of :1306-1322: variables with a finite breakpoint
        const double t = tall[(size_t)rk * (size_t)nmax + (size_t)i];
        if (!(t >= 0.0) || t == inf) continue;
        pv.ht.push_back(t);
        if (pv.h64)
          pv.hio64.push_back(pv.hrow0[rk] + i);
        else
          pv.hio.push_back((uint32_t)(pv.hrow0[rk] + i));
        if (pv.ht.size() == 1 || t < bk) bk = t, pv.hibkmin = (int64_t)pv.ht.size() - 1;
      }
    pv.hleft = (int64_t)pv.ht.size();
    pv.hbuilt = false;
    pv.have = true, pv.full = true;
    pv.win_hi = inf;
    pv.M.clear();
    pv.mpos = pv.safe_end = 0;
    pv.more_anywhere = pv.hleft > 0;
    pv.taken.assign(nranks, 0);
    pv.next_chunk = 1;  // the first record is the scan's minimum itself
    return 0;
  }
  int refill_exact(Provider &pv, const T *x, const T *l, const T *u, const T *g, int head, int col) {
    const int recl = 2 * col + 4;
    const uint32_t chunk_cap = (uint32_t)std::min<size_t>((msg_len - 2) / (size_t)recl, CHUNK_MAX);
    const uint32_t want = std::min<uint32_t>(pv.next_chunk, chunk_cap);
    pv.next_chunk = std::min<uint32_t>(std::max<uint32_t>(pv.next_chunk, 16) * 4, chunk_cap);
    // the next `want` pops of the reference's walk (every rank pops the same replicated heap);
    // each rank gathers the records of the rows it owns, in that order
    std::vector<uint64_t> hk;
    std::vector<uint32_t> hi;
    std::vector<int> owner;
    const int64_t nbreak = (int64_t)pv.ht.size();
    const auto io_at = [&](size_t k) -> int64_t { return pv.h64 ? pv.hio64[k] : (int64_t)pv.hio[k]; };
    while (owner.size() < want && pv.hleft > 0) {
      double tj;
      int64_t grow;
      if (pv.hleft == nbreak) {  // iter == 1 (:1384-1389)
        tj = pv.ht[(size_t)pv.hibkmin], grow = io_at((size_t)pv.hibkmin);
      } else {
        if (!pv.hbuilt) {  // iter == 2: the last entry replaces the used one (:1391-1398)
          if (pv.hibkmin != nbreak - 1) {
            pv.ht[(size_t)pv.hibkmin] = pv.ht[(size_t)nbreak - 1];
            if (pv.h64)
              pv.hio64[(size_t)pv.hibkmin] = pv.hio64[(size_t)nbreak - 1];
            else
              pv.hio[(size_t)pv.hibkmin] = pv.hio[(size_t)nbreak - 1];
          }
        }
        if (pv.h64)
          lbh::hpsolb(pv.hleft, pv.ht.data(), pv.hio64.data(), pv.hbuilt ? 1 : 0);
        else
          lbh::hpsolb(pv.hleft, pv.ht.data(), pv.hio.data(), pv.hbuilt ? 1 : 0);
        pv.hbuilt = true;
        tj = pv.ht[(size_t)pv.hleft - 1], grow = io_at((size_t)pv.hleft - 1);
      }
      pv.hleft--;
      const int rk = (int)(std::upper_bound(pv.hrow0.begin(), pv.hrow0.end(), grow) -
                           pv.hrow0.begin()) - 1;
      owner.push_back(rk);
      if (rk == rank) {
        uint64_t bits;
        std::memcpy(&bits, &tj, 8);
        hk.push_back(bits);
        hi.push_back((uint32_t)(grow - row0));
      }
    }
    const uint32_t len = (uint32_t)owner.size(), own = (uint32_t)hk.size();
    pv.M.clear();
    pv.mpos = pv.safe_end = 0;
    pv.more_anywhere = pv.hleft > 0;
    pv.taken.assign(nranks, 0);
    pv.raw = nullptr;
    if (len == 0) return 0;
    if (own) {
      HIPCHK(hipMemcpyAsync(keys[0], hk.data(), (size_t)own * 8, hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(idx[0], hi.data(), (size_t)own * 4, hipMemcpyHostToDevice, stream));
      lbk::launch_cauchy_gather<T>(q, idx[0], keys[0], own, row0, x, l, u, g, Wc(), head, col, r, d_src(), pend,
                                   d_msg + 2);
    }
    CHK(put_header((double)own, (double)pv.hleft));
    const size_t count = 2 + (size_t)len * recl;
    CHK(exchange(count));  // (also orders the pageable uploads above)
    pv.M.resize(len);
    std::vector<uint32_t> cur(nranks, 0);
    for (uint32_t k = 0; k < len; ++k) {
      const int rk = owner[k];
      const double *rec = h_msg_all + (size_t)rk * count + 2 + (size_t)cur[rk]++ * recl;
      pv.M[k] = MRec{rec[0], (int64_t)rec[1], rk, rec};
    }
    pv.safe_end = len;
    return 0;
  }

  // Several ranks with a communicator: all-gather this chunk of every rank's sorted list and merge
  // the runs ON THE DEVICE -- one stable radix sort on t of <= nranks * chunk keys (ranks own ascending
  // row blocks, so equal t keep global index order), one gather -- so that the host receives ONE
  // (t, global index)-ordered run: no MRec per record, no pairwise merges on the host, and with no
  // pair stored the walk's fast loop runs straight over the records, as on a single rank.  (r02
  // merged on the host: ~ 30 ns per record on every rank against ~ 4 for the walk itself.)
  // merged = false: the buffers could not be allocated; the caller falls back to the host merge.
  int exchange_merged(Provider &pv, size_t count, uint32_t chunk, int recl, bool rawmode, bool &merged) {
    merged = false;
    const size_t S = (size_t)nranks * chunk;
    if (S > mg_slots) {
      const size_t cap = (size_t)nranks * ((msg_len - 2) / 4);  // the longest chunks there are (recl = 4)
      auto F = [](auto *&p) {
        if (p) (void)hipFree(p);
        p = nullptr;
      };
      F(mg_keys[0]), F(mg_keys[1]), F(mg_vals[0]), F(mg_vals[1]), F(mg_tmp), F(d_merged);
      mg_slots = 0;
      mg_tmp_bytes = lbk::sort_pairs_temp_bytes(cap) + 256;
      bool ok = hipMalloc(&mg_keys[0], cap * 8) == hipSuccess && hipMalloc(&mg_keys[1], cap * 8) == hipSuccess &&
                hipMalloc(&mg_vals[0], cap * 4) == hipSuccess && hipMalloc(&mg_vals[1], cap * 4) == hipSuccess &&
                hipMalloc(&mg_tmp, mg_tmp_bytes) == hipSuccess &&
                hipMalloc(&d_merged, (size_t)nranks * mg_stride() * sizeof(double)) == hipSuccess;
      if (!ok) {
        (void)hipGetLastError();
        F(mg_keys[0]), F(mg_keys[1]), F(mg_vals[0]), F(mg_vals[1]), F(mg_tmp), F(d_merged);
        // (every rank must take the same route: without the buffers HERE the run cannot go on in step)
        return fail(LBFGSB_E_ALLOC, "no memory for the merge buffers of the breakpoint exchange");
      }
      mg_slots = cap;
    }
    ncoll++, coll_bytes += (int64_t)count * 8;
    if (g_rccl.AllGather(d_msg, d_msg_all, count, ncclDouble, comm, stream) != ncclSuccess)
      return fail(LBFGSB_E_COMM, "ncclAllGather failed");
    lbk::launch_merge_chunks(q, nranks, chunk, recl, count, d_msg_all, mg_keys[0], mg_keys[1], mg_vals[0],
                             mg_vals[1], mg_tmp, mg_tmp_bytes, d_merged);
    const size_t out_doubles = 4 * (size_t)nranks + S * (size_t)recl + (S + 7) / 8;
    HIPCHK(hipMemcpyAsync(h_msg_all, d_merged, out_doubles * sizeof(double), hipMemcpyDeviceToHost, stream));
    {
      const double t0 = now_s();
      HIPCHK(hipStreamSynchronize(stream));
      t_wait += now_s() - t0;
    }
    nsync++;
    pf_valid = false;
    const double *hdr = h_msg_all, *recs = h_msg_all + 4 * (size_t)nranks;
    const unsigned char *rb = reinterpret_cast<const unsigned char *>(recs + S * (size_t)recl);
    size_t total = 0;
    pv.more_anywhere = false;
    double bt = std::numeric_limits<double>::infinity();
    int64_t bi = std::numeric_limits<int64_t>::max();
    for (int rk = 0; rk < nranks; ++rk) {
      const uint32_t lr = (uint32_t)hdr[4 * rk];
      total += lr;
      if (hdr[4 * rk + 1] > 0.0) {  // this rank holds later records: nothing beyond its last one is safe
        pv.more_anywhere = true;
        const double lt = hdr[4 * rk + 2];
        const int64_t li = (int64_t)hdr[4 * rk + 3];
        if (lt < bt || (lt == bt && li < bi)) bt = lt, bi = li;
      }
    }
    pv.M.clear();
    pv.raw = nullptr, pv.rank_of = rb;
    if (rawmode) {
      pv.raw = recs, pv.raw_n = total;
    } else {
      pv.M.resize(total);
      for (size_t k = 0; k < total; ++k) {
        const double *rec = recs + k * (size_t)recl;
        pv.M[k] = MRec{rec[0], (int64_t)rec[1], (int)rb[k], rec};
      }
    }
    pv.safe_end = total;
    if (pv.more_anywhere) {  // first record after (bt, bi): everything before it is safe to consume
      size_t lo = 0, hi = total;
      while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const double *rec = recs + mid * (size_t)recl;
        const bool le = rec[0] < bt || (rec[0] == bt && (int64_t)rec[1] <= bi);
        if (le) lo = mid + 1; else hi = mid;
      }
      pv.safe_end = lo;
    }
    pv.mpos = 0;
    pv.taken.assign(nranks, 0);
    merged = true;
    return 0;
  }

  // all-gather the next chunk of every rank's local list and merge
  int refill(Provider &pv, const T *x, const T *l, const T *u, const T *g, int head, int col) {
    if (pv.exact) return refill_exact(pv, x, l, u, g, head, col);
    const int recl = 2 * col + 4;
    const uint32_t chunk = pv.next_chunk;
    // the message buffer holds CHUNK_MAX records of the widest kind (col = m); narrower records
    // (col = 0 on the first iteration: 4 doubles) travel in proportionally longer chunks
    const uint32_t chunk_cap = (uint32_t)((msg_len - 2) / (size_t)recl);
    pv.next_chunk = std::min<uint32_t>(pv.next_chunk * 4, chunk_cap);
    const uint32_t len = std::min<uint32_t>(chunk, pv.Cl - pv.pl);
    const size_t count = 2 + (size_t)chunk * recl;
    const bool single = nranks == 1 && !comm;
    if (single && pf_valid && pf_pl == pv.pl && pf_len == len && pf_cur == pv.cur) {
      // this chunk was gathered and copied while the host walked the previous one
      const double t0 = now_s();
      HIPCHK(hipEventSynchronize(pf_ev));
      t_wait += now_s() - t0;
      nsync++;
      std::swap(h_msg_all, h_msg_loc);
      std::swap(d_msg, d_msg2);
      h_msg_all[0] = (double)pf_len, h_msg_all[1] = (double)pf_rem;
    } else {
      lbk::launch_cauchy_gather<T>(q, idx[pv.cur] + pv.pl, keys[pv.cur] + pv.pl, len, row0, x, l, u, g,
                                   Wc(), head, col, r, d_src(), pend, d_msg + 2);
      CHK(put_header((double)len, (double)(pv.Cl - pv.pl - len)));
      if (comm && nranks > 1 && nranks <= 255 && !debug_walk) {
        bool merged = false;
        CHK(exchange_merged(pv, count, chunk, recl, col == 0 && print_level < 100, merged));
        if (merged) return 0;
      }
      CHK(exchange(count));
    }
    pf_valid = false;
    const bool rawmode = single && col == 0 && print_level < 100 && !debug_walk;
    pv.raw = nullptr;
    pv.rank_of = nullptr;
    if (single && pv.Cl - pv.pl > len) {  // prefetch the chunk after this one
      const uint32_t npl = pv.pl + len;
      const uint32_t nlen = std::min<uint32_t>(pv.next_chunk, pv.Cl - npl);
      lbk::launch_cauchy_gather<T>(q, idx[pv.cur] + npl, keys[pv.cur] + npl, nlen, row0, x, l, u, g, Wc(),
                                   head, col, r, d_src(), pend, d_msg2 + 2);
      HIPCHK(hipMemcpyAsync(h_msg_loc, d_msg2, (2 + (size_t)nlen * recl) * sizeof(double),
                            hipMemcpyDeviceToHost, stream));
      HIPCHK(hipEventRecord(pf_ev, stream));
      pf_valid = true, pf_pl = npl, pf_len = nlen, pf_rem = pv.Cl - npl - nlen, pf_cur = pv.cur;
    }
    pv.M.clear();
    pv.more_anywhere = false;
    double bt = std::numeric_limits<double>::infinity();
    int64_t bi = std::numeric_limits<int64_t>::max();
    for (int rk = 0; rk < nranks; ++rk) {
      const double *base = h_msg_all + (size_t)rk * count;
      const uint32_t lr = (uint32_t)base[0];
      if (rawmode) {  // (no MRec per record: sizing M would write 32 bytes for each of them)
        pv.raw = base + 2;
        pv.raw_n = lr;
      } else {
        const size_t at = pv.M.size();
        pv.M.resize(at + lr);
        MRec *out = pv.M.data() + at;
        for (uint32_t k = 0; k < lr; ++k) {
          const double *rec = base + 2 + (size_t)k * recl;
          out[k] = MRec{rec[0], (int64_t)rec[1], rk, rec};
        }
      }
      if (base[1] > 0.0) {  // this rank holds later records: nothing beyond its last one is safe
        pv.more_anywhere = true;
        const double *last = base + 2 + (size_t)(lr - 1) * recl;
        if (last[0] < bt || (last[0] == bt && (int64_t)last[1] < bi)) bt = last[0], bi = (int64_t)last[1];
      }
    }
    auto less = [](const MRec &a, const MRec &b) {
      return a.t < b.t || (a.t == b.t && a.gidx < b.gidx);
    };
    if (nranks > 1) {
      // every rank's run is already sorted: merge the runs pairwise (O(N log ranks))
      std::vector<size_t> cut;
      cut.push_back(0);
      for (size_t k = 1; k < pv.M.size(); ++k)
        if (pv.M[k].rank != pv.M[k - 1].rank) cut.push_back(k);
      cut.push_back(pv.M.size());
      while (cut.size() > 2) {
        std::vector<size_t> nxt;
        for (size_t k = 0; k + 2 < cut.size(); k += 2) {
          std::inplace_merge(pv.M.begin() + cut[k], pv.M.begin() + cut[k + 1],
                             pv.M.begin() + cut[k + 2], less);
          nxt.push_back(cut[k]);
        }
        if (cut.size() % 2 == 0) nxt.push_back(cut[cut.size() - 2]);
        nxt.push_back(pv.M.size());
        cut.swap(nxt);
      }
    }
    pv.safe_end = pv.msize();
    if (pv.more_anywhere && nranks > 1) {  // (a single rank's own run is safe to its end)
      size_t k = 0;
      while (k < pv.M.size() && (pv.M[k].t < bt || (pv.M[k].t == bt && pv.M[k].gidx <= bi))) ++k;
      pv.safe_end = k;
    }
    pv.mpos = 0;
    pv.taken.assign(nranks, 0);
    if (debug_walk) {
      std::fprintf(stderr, "[refill] chunk=%u len=%u Cl=%u pl=%u cur=%d M=%zu safe=%zu more=%d\n", chunk,
                   len, pv.Cl, pv.pl, pv.cur, pv.M.size(), pv.safe_end, (int)pv.more_anywhere);
      for (size_t k = 0; k < pv.M.size() && k < 30; ++k)
        std::fprintf(stderr, "   rec %zu: t=%.17g gidx=%lld d=%g z=%g\n", k, pv.M[k].t,
                     (long long)pv.M[k].gidx, pv.M[k].rec[2], pv.M[k].rec[3]);
    }
    return 0;
  }
