// solver_cauchy.inl -- member functions of Solver<T> (included inside the class body in solver.hip):
// the generalized Cauchy point (reference src/lbfgsb.f90:1157-1532).  The n-loop runs on the device
// (cauchy_scan_kernel, or fused into update_scan_kernel); this file holds the breakpoint PROVIDER --
// window compaction, (t, index) ordering, chunked record gathers, the all-gather + merge over ranks,
// the reference's own heap order for walks that end inside a group of equal breakpoints -- the exact
// host replay of the walk (:1378-1497), and the opt-in parallel search (LBFGSB_F_PARALLEL_GCP).
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
  int ensure_d(const T *x) {
    if (!d_impl) return 0;
    lbk::launch_dz_materialise<T>(q, n, x, t, d, z_in_x ? z : (T *)nullptr);
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
    double res[lbk::RES_MAX];
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
    lbk::launch_cauchy_gather_dyn<T>(q, sp_idx, sp_keys, sp_count, SPEC_CAP, row0, x, l, u, g, W(), head,
                                     col, r, d_src(), lbk::Pend{1, stp, d_impl ? 1 : 0}, sp_msg);
    const size_t cnt = 2 + (size_t)SPEC_CAP * (2 * col + 4);
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
    lbk::launch_cauchy_gather_dyn<T>(q, idx[0], keys[0], d_count, FAST_CAP, row0, x, l, u, g, W(),
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
      lbk::launch_cauchy_gather<T>(q, idx[0], keys[0], own, row0, x, l, u, g, W(), head, col, r, d_src(), pend,
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
                                   W(), head, col, r, d_src(), pend, d_msg + 2);
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
      lbk::launch_cauchy_gather<T>(q, idx[pv.cur] + npl, keys[pv.cur] + npl, nlen, row0, x, l, u, g, W(),
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

  // The Cauchy point is kept in functional form (tsum + iwhere, see xcp_row in kernels_common.hpp)
  // and only written out as a vector where one is needed: subsm skipped, the backtracking
  // branch of subsm, state export.
  struct Gcp {
    double tsum = 0.0, last_t = -1.0;
    int64_t last_i = -1;
    bool copy_x = false;  // xcp = x without a cauchy scan behind it (tbrk is stale)
  } gcp;
  bool z_valid = false;
  static constexpr size_t FIX_CAP = 65536;
  std::vector<int64_t> fixlist;
  bool fix_overflow = false;
  int64_t *d_fix = nullptr, *h_fix = nullptr;

  int write_xcp(T *dst, const T *x, const T *l, const T *u, const T *g) {
    if (gcp.copy_x) {
      HIPCHK(hipMemcpyAsync(dst, x, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    } else {
      lbk::launch_xcp_fill<T>(q, n, x, g, l, u, iwhere, gcp.tsum, dst);
    }
    return 0;
  }
  int ensure_z(const T *x, const T *l, const T *u, const T *g) {
    if (!z_valid) CHK(write_xcp(z, x, l, u, g));
    z_valid = true;
    return 0;
  }
  // end of cauchy: make iwhere final (rows fixed by the walk) without writing xcp
  int close_gcp(double tsum, double last_t, int64_t last_i) {
    gcp.tsum = tsum, gcp.last_t = last_t, gcp.last_i = last_i, gcp.copy_x = false;
    z_valid = false;
    iw_dirty += fix_overflow ? 1.0 : (double)fixlist.size();  // rows whose iwhere the walk sets
    return apply_walk_fixes();
  }
  // (also: a deferred line-search set-up that has to redo subsm's backtracking branch puts the walk's
  //  fixes back on top of the recomputed post-scan status, solver.hip land_deferred)
  int apply_walk_fixes() {
    if (gcp.copy_x) return 0;  // (no walk behind this Cauchy point)
    if (fix_overflow) {  // long walk: the cursor-based kernel (it writes z on the way)
      CHK(ensure_tbrk());
      lbk::launch_cauchy_finish<T>(q, n, row0, (const T *)cx, (const T *)cl, (const T *)cu,
                                   (const T *)cg, tbrk, iwhere, z, gcp.tsum, gcp.last_t, gcp.last_i);
      z_valid = true;
    } else {
      for (size_t at = 0; at < fixlist.size(); at += FIX_CAP) {  // (one piece unless exact order)
        const size_t cnt = std::min(FIX_CAP, fixlist.size() - at);
        if (at) HIPCHK(hipStreamSynchronize(stream));  // h_fix is reused
        std::memcpy(h_fix, fixlist.data() + at, cnt * sizeof(int64_t));
        HIPCHK(hipMemcpyAsync(d_fix, h_fix, cnt * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        lbk::launch_cauchy_fix(q, d_fix, (int)cnt, row0, n, iwhere);
      }
    }
    return 0;
  }
  const void *cx = nullptr, *cl = nullptr, *cu = nullptr, *cg = nullptr;  // this call's operands

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
      lbk::launch_pgcp_gather<T>(q, idx[1], keys[1], nb, nbp, x, l, u, g, W(), head, col, theta, r,
                                 d_src(), pend, tt, dd, a0, wb, pp, (double *)nullptr, row0);
    } else {
      // own records in local order -> all ranks -> merged by (t, global index): the merge sort is
      // stable and ranks own ascending row blocks, so equal t keep global index order
      double *Lt = L, *Ld = L + lbp, *La = L + 2 * lbp, *Lg = L + 3 * lbp, *Lw = L + 4 * lbp,
             *Lu = Lw + (size_t)col2 * lbp;
      HIPCHK(hipMemsetAsync(L, 0, (size_t)narr_l * lbp * sizeof(double), stream));
      if (cnt)
        lbk::launch_pgcp_gather<T>(q, idx[1], keys[1], (int64_t)cnt, lbp, x, l, u, g, W(), head, col, theta,
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
  void dump_cauchy_x(const T *x, const T *l, const T *u, const T *g) {  // :1345, :1527
    (void)write_xcp(xp, x, l, u, g);
    const std::vector<double> v = host_vec(xp);
    rep.vec_rows("Cauchy X =  ", v.data(), n);
  }

  // Generalized Cauchy point, reference :1157-1532.  p,c,wbp,v = wa8m slots.
  // results of the n-loop of cauchy when it was fused into the matupd pass
  struct ScanOut {
    bool ready = false;
    std::vector<double> p;  // 2 m
    double f1 = 0, nbreak = 0, nunb = 0, nunbnz = 0, bkmin = 0;
  } scan;

  // The long stretch of a first-iteration walk (col = 0, records of 4 doubles: t, row, d, z, in order): the
  // reference's steps :1416-1434, :1452-1453, :1483-1497 in its operation order, on values that live in
  // registers, the records prefetched ahead.  -> 0: records used up; 1: the walk stops here; 2: all n variables
  // fixed.  tie: the walk stops at a breakpoint equal to the last one it crossed.
  struct WalkRaw {
    double f1, f2, dtm, tsum, tj, lt;
    int64_t nleft;
  };
  static __attribute__((noinline)) int walk_raw_col0(const double *raw, size_t &pos_io, size_t end, double theta,
                                                     double clampv, bool all_n, bool bnded, WalkRaw &w,
                                                     bool &tie) {
    const double INFL = 1.0 + 16.0 * std::numeric_limits<double>::epsilon();
    const double inf = std::numeric_limits<double>::infinity();
    double f1_ = w.f1, f2_ = w.f2, dtm_ = w.dtm, tsum_ = w.tsum, tj_ = w.tj, lt_ = w.lt;
    int64_t nleft_ = w.nleft;
    size_t pos = pos_io;
    int code = 0;
    bool tie_ = false;
    while (pos < end) {
      const double *rec = raw + pos * 4;
      __builtin_prefetch(rec + 96);
      const double mt = rec[0];
      if (!(mt <= (tj_ + dtm_) * INFL && mt < inf)) {  // beyond reach: dtm < dt
        tie_ = lt_ >= 0.0 && mt == lt_;
        code = 1;
        break;
      }
      const double dt = mt - tj_;
      if (dtm_ < dt) {  // :1416
        tie_ = lt_ >= 0.0 && mt == lt_;
        code = 1;
        break;
      }
      ++pos;
      tsum_ = tsum_ + dt;
      nleft_ = nleft_ - 1;
      const double dibp = rec[2];
      const double zibp = rec[3];
      tj_ = mt;
      lt_ = mt;
      if (nleft_ == 0 && all_n) {  // all n variables fixed (:1436-1442)
        dtm_ = dt;
        code = 2;
        break;
      }
      const double dibp2 = dibp * dibp;
      f1_ = f1_ + dt * f2_ + dibp2 - theta * dibp * zibp;  // :1452-1453
      f2_ = f2_ - theta * dibp2;
      f2_ = std::max(clampv, f2_);  // :1483
      if (nleft_ > 0) {
        dtm_ = -f1_ / f2_;
      } else if (bnded) {
        f1_ = 0.0, f2_ = 0.0, dtm_ = 0.0;
        code = 1;
        break;
      } else {
        dtm_ = -f1_ / f2_;
        code = 1;
        break;
      }
    }
    w = WalkRaw{f1_, f2_, dtm_, tsum_, tj_, lt_, nleft_};
    pos_io = pos;
    tie = tie_;
    return code;
  }

  int cauchy(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, double theta,
             int col, int head, double sbgnrm, double epsmch, int &nseg, int &info) {
    double *p = &wa8m[0], *c = &wa8m[2 * m], *wbp = &wa8m[4 * m], *v = &wa8m[6 * m];
    cx = x, cl = l, cu = u, cg = g, cnbd = nbd;
    pf_valid = false;
    fixlist.clear();
    fix_overflow = false;
    closed_ok = false;
    z_in_x = false;  // z means this call's Cauchy point from here on
    std::memset(nrc, 0, sizeof nrc);
    const int ipr = quiet ? -1 : print_level;
    if (sbgnrm <= 0.0) {  // :1245-1249
      scan.ready = false;
      gcp = Gcp{};
      gcp.copy_x = true;
      z_valid = false;
      return 0;
    }
    const int col2 = 2 * col;
    const int MC = col ? lbk::maxc_for(col) : 0;
    if (ipr >= 99) std::fprintf(rep.out, "\n---------------- CAUCHY entered-------------------\n");
    auto leave = [&](double tsum_, double lt, int64_t li) -> int {  // update() :1519-1530
      CHK(close_gcp(tsum_, lt, li));
      if (ipr > 100) dump_cauchy_x(x, l, u, g);
      if (ipr >= 99) std::fprintf(rep.out, "\n---------------- exit CAUCHY----------------------\n\n");
      return 0;
    };
    if (!scan.ready && wide()) {
      CHK(wide_cauchy_scan(x, l, u, nbd, g, head, col));
    } else if (!scan.ready) {
      lbk::launch_cauchy_scan<T>(q, n, x, l, u, nbd, g, iwhere, tbrk, W(), head, col);
      iw_dirty += 1.0;  // (this scan does not count the entries it changes)
      tbrk_valid = true;
      CHK(fetch(2 * MC + 4, 1, 0));
      for (int j = 0; j < col; ++j) {
        scan.p[j] = h_res[j];
        scan.p[col + j] = h_res[MC + j];
      }
      scan.f1 = h_res[2 * MC], scan.nbreak = h_res[2 * MC + 1], scan.nunb = h_res[2 * MC + 2];
      scan.nunbnz = h_res[2 * MC + 3], scan.bkmin = h_res[2 * MC + 4];
    }
    scan.ready = false;
    for (int j = 0; j < col2; ++j) p[j] = scan.p[j];
    double f1 = scan.f1;
    const int64_t nbreak = (int64_t)scan.nbreak;
    const int64_t nunb = (int64_t)scan.nunb;
    const bool bnded = scan.nunbnz == 0.0;
    const double bkmin = scan.bkmin;
    if (theta != 1.0)
      for (int j = 0; j < col; ++j) p[col + j] = theta * p[col + j];  // :1337
    p_ini_max = 0.0;
    for (int j = 0; j < 2 * col; ++j) p_ini_max = std::max(p_ini_max, std::fabs(p[j]));

    double last_t = -1.0;
    int64_t last_i = -1;
    if (nbreak == 0 && nunb == 0) {  // d = 0: xcp = x (:1343-1347)
      CHK(close_gcp(0.0, last_t, last_i));
      if (ipr > 100) dump_cauchy_x(x, l, u, g);
      return 0;
    }
    for (int j = 0; j < col2; ++j) c[j] = 0.0;
    double f2 = -theta * f1;  // :1357-1363
    const double f2_org = f2;
    if (col > 0) {
      info = lbh::bmv(m, sy.data(), wt.data(), col, p, v);
      if (info != 0) return 0;
      f2 = f2 - lbh::dot_seq(col2, v, p);
    }
    double dtm = -f1 / f2;
    double tsum = 0.0;
    nseg = 1;
    last_dtm0 = dtm;
    if (ipr >= 99) std::fprintf(rep.out, " There are %11lld   breakpoints \n", (long long)nbreak);  // :1367

    if (col == 0 && nbreak != 0 && (flags & LBFGSB_F_PARALLEL_GCP) && dtm >= bkmin) {
      // B = theta*I: phi'(t) = -(1 - theta t) * (remaining d'd), so the walk stops at t = 1/theta
      // having fixed exactly the breakpoints t_j <= 1/theta (see include/lbfgsb_hip.h).
      const double tstar = 1.0 / theta;
      // ... as long as the reference's clamp f2 = max(epsmch f2_org, f2) (:1483) cannot act before
      // t*: f2 = theta * (d'd over the rows still moving), which only shrinks along the walk, so
      // it is enough to look at what is left beyond t* (with a margin for the rounding noise the
      // sequential recurrence carries); otherwise: the exact replay below
      CHK(ensure_tbrk());
      lbk::launch_gcp_rest_mass<T>(q, n, g, tbrk, tstar);
      CHK(fetch(1, 0, 0));
      if (h_res[0] >= 1.0e4 * epsmch * (-f1)) {
      lbk::launch_cauchy_finish<T>(q, n, row0, x, l, u, g, tbrk, iwhere, z, tstar, tstar,
                                   std::numeric_limits<int64_t>::max(), 1);
      iw_dirty += 1.0;
      gcp = Gcp{};
      gcp.tsum = tstar, gcp.last_t = tstar, gcp.last_i = std::numeric_limits<int64_t>::max();
      z_valid = true;
      CHK(fetch(1, 0, 0));
      const int64_t done = (int64_t)h_res[0];
      // the walk counts a segment per fixed variable except a last one that fixes all n (:1436)
      const int64_t ns = 1 + done - ((done == nbreak && nbreak == nglob) ? 1 : 0);
      nseg = (int)std::min<int64_t>(ns, std::numeric_limits<int>::max());
      if (ipr >= 99) std::fprintf(rep.out, "\n---------------- exit CAUCHY----------------------\n\n");
      return 0;
      }
      ngcp_clamped++;
    }

    // Equal breakpoints are delivered in index order, the reference pops them in heap order
    // (hpsolb :2079); the two differ in effect only if the walk ends INSIDE such a group.  That
    // is detected (tie_split), counted, and the walk is then replayed from its start in the
    // reference's own order (exact_init / refill_exact) -- unless LBFGSB_F_INDEX_TIES opts out.
    const bool can_exact = !(flags & LBFGSB_F_INDEX_TIES);
    // (a replay would print the walk twice: under iprint >= 99 the walk runs in that order from the
    //  start; option "exact_always": every walk in that order, for tests)
    bool exact_run = can_exact && (print_level >= 99 || exact_always);
    std::vector<double> p_start(p, p + col2);
    const double f1_start = f1, f2_start = f2, dtm_start = dtm;
    for (;;) {  // at most two trips: the second one in exact order
    bool tie_split = false;
    if (nbreak != 0) {
      int64_t nleft = nbreak;
      int64_t iter = 1;
      double tj = 0.0;
      Provider pv;
      if (exact_run) CHK(exact_init(pv));
      const double INFL = 1.0 + 16.0 * std::numeric_limits<double>::epsilon();
      for (;;) {
        const double tj0 = tj;
        // (control flow follows print_level, which every rank shares -- ipr is -1 on the quiet ranks)
        if (iter == 1 && print_level < 100) {  // smallest breakpoint known from the scan: usual exit (:1384-1389)
          if (dtm < bkmin - tj0) break;
        }
        // ---- no pair stored and records on the host: the same steps as below in a tight loop
        //      (the first iteration walks ~n of them; per record only :1416-1434, :1452-1453,
        //       :1483-1497 remain, in the reference's operation order) ----
        if (col == 0 && print_level < 100 && pv.have && pv.mpos < pv.safe_end) {
          const MRec *M = pv.M.data();
          size_t pos = pv.mpos;
          const size_t end = pv.safe_end;
          const double inf = std::numeric_limits<double>::infinity();
          bool stop = false;
          if (pv.raw && fix_overflow && !pv.exact && pos < end) {
            // The long stretch of a first-iteration walk (single rank, > 65 536 segments behind it: the
            // rows it fixes are described by a cursor, not a list): the same operations in the same
            // order on LOCAL copies of the walk's state -- nothing in the loop can alias them, so they
            // stay in registers -- with the records prefetched ahead (they were written by DMA: every
            // line is a cache miss, and the branch on dtm keeps the hardware prefetcher from running
            // ahead).  1.5 - 2 x the rate of the general loop below (profiles/scripts/walk_bench.cpp).
            // (a function of its own, NOT inlined: inside this long routine the loop's seven running
            //  values were spilled to the stack -- a store-to-load round trip on the chain f1 -> f1 of every
            //  segment, 3.3 ns per record; on its own it keeps them in registers: 1.7 ns, the rate of
            //  profiles/scripts/walk_bench.cpp)
            WalkRaw w{f1, f2, dtm, tsum, tj, last_t, nleft};
            const size_t pos0 = pos;
            bool tie_ = false;
            const int code = walk_raw_col0(pv.raw, pos, end, theta, epsmch * f2_org, nbreak == nglob, bnded, w, tie_);
            const double f1_ = w.f1, f2_ = w.f2, dtm_ = w.dtm, tsum_ = w.tsum, tj_ = w.tj, lt_ = w.lt;
            const int64_t nleft_ = w.nleft;
            const double *const raw = pv.raw;
            const int64_t took = (int64_t)(pos - pos0);
            f1 = f1_, f2 = f2_, dtm = dtm_, tsum = tsum_, tj = tj_, last_t = lt_, nleft = nleft_;
            iter += took;
            if (pv.rank_of)
              for (size_t k = pos0; k < pos; ++k) pv.taken[pv.rank_of[k]]++;
            else
              pv.taken[0] += (uint32_t)took;
            if (took > 0) last_i = (int64_t)raw[(pos - 1) * 4 + 1];
            nseg += (int)(code == 2 ? took - 1 : took);  // (the all-fixed exit does not count its segment)
            pv.mpos = pos;
            if (code == 2) return leave(tsum, last_t, last_i);
            if (code == 1) {
              tie_split = tie_;
              break;
            }
            continue;  // records used up: refill below on the next trip
          }
          while (pos < end) {
            // (single rank: the records themselves, 4 doubles each, in order; else the merged list)
            const double *rec = pv.raw ? pv.raw + pos * 4 : M[pos].rec;
            const double mt = rec[0];
            if (!(mt <= (tj + dtm) * INFL && mt < inf)) {  // beyond reach: dtm < dt
              tie_split = last_t >= 0.0 && mt == last_t;
              stop = true;
              break;
            }
            const double dt = mt - tj;
            if (dtm < dt) {  // :1416
              tie_split = last_t >= 0.0 && mt == last_t;
              stop = true;
              break;
            }
            pv.taken[pv.raw ? (pv.rank_of ? (int)pv.rank_of[pos] : 0) : M[pos].rank]++;
            ++pos;
            tsum = tsum + dt;
            nleft = nleft - 1;
            iter = iter + 1;
            const double dibp = rec[2];
            const double zibp = rec[3];
            tj = mt;
            last_t = mt;
            last_i = (int64_t)rec[1];
            if (!fix_overflow) {
              if (pv.exact || fixlist.size() < FIX_CAP)  // (exact order: no cursor describes the set)
                fixlist.push_back(last_i * 2 + (dibp > 0.0 ? 1 : 0));
              else
                fix_overflow = true;
            }
            if (nleft == 0 && nbreak == nglob) {  // all n variables fixed (:1436-1442)
              dtm = dt;
              pv.mpos = pos;
              return leave(tsum, last_t, last_i);
            }
            nseg = nseg + 1;
            const double dibp2 = dibp * dibp;
            f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;  // :1452-1453
            f2 = f2 - theta * dibp2;
            f2 = std::max(epsmch * f2_org, f2);  // :1483
            if (nleft > 0) {
              dtm = -f1 / f2;
            } else if (bnded) {
              f1 = 0.0;
              f2 = 0.0;
              dtm = 0.0;
              stop = true;
              break;
            } else {
              dtm = -f1 / f2;
              stop = true;
              break;
            }
          }
          pv.mpos = pos;
          if (stop) break;
          continue;  // records used up: refill below on the next trip
        }
        // ---- next breakpoint after (last_t, last_i), if it can matter: t <= tj0 + dtm ----
        // (iprint >= 100 reports the distance to the next breakpoint of every segment, :1408-1412:
        //  then the next one is always fetched)
        const double hi_need =
            print_level >= 100 ? std::numeric_limits<double>::infinity() : (tj0 + dtm) * INFL;
        const double *rec = nullptr;
        int64_t rec_gi = -1;
        bool to_tight_loop = false;
        for (;;) {
          if (pv.raw && pv.have && pv.mpos < pv.safe_end) {  // (M is not filled in this mode)
            to_tight_loop = true;
            break;
          }
          if (pv.have && pv.mpos < pv.safe_end) {
            const MRec &mr = pv.M[pv.mpos];
            if (mr.t <= hi_need && mr.t < std::numeric_limits<double>::infinity()) {
              rec = mr.rec;
              rec_gi = mr.gidx;
            }
            break;
          }
          if (pv.have && (pv.mpos < pv.msize() || pv.more_anywhere)) {
            pv.pl += pv.taken.empty() ? 0 : pv.taken[rank];
            CHK(refill(pv, x, l, u, g, head, col));
            continue;
          }
          if (pv.have && (pv.full || pv.win_hi >= hi_need)) break;  // nothing left in reach
          // (re)fetch: ask further ahead each time so long walks need few round trips
          double hi = hi_need;
          if (pv.grow > 0 && std::isfinite(hi_need)) {
            const double base = last_t > 0 ? last_t : 0.0;
            hi = base + (hi_need - base) * std::ldexp(1.0, std::min(pv.grow, 40));
          }
          pv.grow++;
          double in_window = 0.0;
          const bool may_pg = col > 0 && (flags & LBFGSB_F_PARALLEL_GCP) && iter == 1 && !pv.have &&
                              print_level < 99 && !wide();
          CHK(window_fetch(pv, last_t, last_i, hi, x, l, u, g, head, col, may_pg ? &in_window : nullptr));
          if (may_pg && in_window > PG_MIN) {
            // many breakpoints within reach and pairs stored: sort + scans on the device (opt-in)
            bool done = false;
            CHK(parallel_gcp(x, l, u, g, theta, col, head, p, c, f1, f2, f2_org, bnded, nbreak, nseg, info,
                             done));
            if (info != 0) return 0;
            if (done) return 0;
            pv.grow = 0;  // (did not fit in memory: replay the walk as usual)
            pv.have = false;
            CHK(window_fetch(pv, last_t, last_i, hi, x, l, u, g, head, col));
          }
        }
        if (to_tight_loop) continue;
        if (!rec) {  // next breakpoint is beyond tj0 + dtm  =>  dtm < dt
          tie_split = last_t >= 0.0 && pv.have && pv.mpos < pv.safe_end && pv.M[pv.mpos].t == last_t;
          break;
        }
        tj = rec[0];
        const double dt = tj - tj0;
        if (dt != 0.0 && ipr >= 100) {  // :1408-1412
          std::fprintf(rep.out, "\n");
          rep.piece(nseg, f1, f2);
          std::fprintf(rep.out, "Distance to the next break point =  %s\n", lbr::fD(dt, 11, 4).c_str());
          std::fprintf(rep.out, "Distance to the stationary point =  %s\n", lbr::fD(dtm, 11, 4).c_str());
        }
        if (dtm < dt) {  // :1416
          tie_split = last_t >= 0.0 && tj == last_t;
          break;
        }

        // fix this variable (:1421-1434)
        pv.taken[pv.M[pv.mpos].rank]++;
        pv.mpos++;
        tsum = tsum + dt;
        nleft = nleft - 1;
        iter = iter + 1;
        const double dibp = rec[2];
        const double zibp = rec[3];
        last_t = tj;
        last_i = rec_gi;
        if (pv.exact || fixlist.size() < FIX_CAP)
          fixlist.push_back(rec_gi * 2 + (dibp > 0.0 ? 1 : 0));
        else
          fix_overflow = true;
        if (col > 0 && col <= two_pass_maxcol) {
          // this row leaves the free set: its share of formk's new row/column moves from the
          // free sums to the active ones (the update pass summed with the pre-walk split)
          const double yk = rec[4 + col - 1], sk = rec[4 + 2 * col - 1];
          for (int j = 0; j < col; ++j) {
            nrc[0][j] += yk * rec[4 + j];        // - sum_free y_new Wy_j
            nrc[1][j] += sk * rec[4 + col + j];  // + sum_act  s_new Ws_j
            nrc[2][j] += sk * rec[4 + j];        // + sum_act  s_new Wy_j
            nrc[3][j] += rec[4 + col + j] * yk;  // - sum_free Ws_j y_new
          }
        }
        if (ipr >= 100)  // :1435
          std::fprintf(rep.out, " Variable  %11lld   is fixed.\n", (long long)rec_gi + 1);
        if (nleft == 0 && nbreak == nglob) {  // all n variables fixed (:1436-1442)
          dtm = dt;
          if (col > 0)
            for (int j = 0; j < col2; ++j) c[j] = c[j] + dtm * p[j];
          return leave(tsum, last_t, last_i);  // no row is left to move: tsum is moot
        }
        nseg = nseg + 1;
        const double dibp2 = dibp * dibp;
        f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;  // :1452-1453
        f2 = f2 - theta * dibp2;
        if (col > 0) {
          if (dt != 0.0)
            for (int j = 0; j < col2; ++j) c[j] = c[j] + dt * p[j];
          for (int j = 0; j < col; ++j) {
            wbp[j] = rec[4 + j];
            wbp[col + j] = theta * rec[4 + col + j];
          }
          info = lbh::bmv(m, sy.data(), wt.data(), col, wbp, v);
          if (info != 0) return 0;
          const double wmc = lbh::dot_seq(col2, c, v);
          const double wmp = lbh::dot_seq(col2, p, v);
          const double wmw = lbh::dot_seq(col2, wbp, v);
          if (-dibp != 0.0)
            for (int j = 0; j < col2; ++j) p[j] = p[j] + (-dibp) * wbp[j];
          f1 = f1 + dibp * wmc;
          f2 = f2 + 2.0 * dibp * wmp - dibp2 * wmw;
        }
        f2 = std::max(epsmch * f2_org, f2);  // :1483
        if (nleft > 0) {
          dtm = -f1 / f2;
        } else if (bnded) {
          f1 = 0.0;
          f2 = 0.0;
          dtm = 0.0;
          break;
        } else {
          dtm = -f1 / f2;
          break;
        }
      }
    }
    if (tie_split && !exact_run) {
      ntiesplit++;
      if (can_exact) {  // replay from the start of the walk, in the reference's order
        exact_run = true;
        std::copy(p_start.begin(), p_start.end(), p);
        for (int j = 0; j < col2; ++j) c[j] = 0.0;
        f1 = f1_start, f2 = f2_start, dtm = dtm_start, tsum = 0.0, nseg = 1;
        last_t = -1.0, last_i = -1;
        fixlist.clear();
        fix_overflow = false;
        std::memset(nrc, 0, sizeof nrc);
        continue;
      }
    }
    break;
    }
    if (debug_walk)
      std::fprintf(stderr, "[cauchy] nseg=%d tsum=%g dtm=%g last=(%.17g,%lld)\n", nseg, tsum, dtm,
                   last_t, (long long)last_i);
    if (ipr >= 99) {  // :1502-1508
      std::fprintf(rep.out, "\n GCP found in this segment\n");
      rep.piece(nseg, f1, f2);
      std::fprintf(rep.out, "Distance to the stationary point =  %s\n", lbr::fD(dtm, 11, 4).c_str());
    }
    if (dtm <= 0.0) dtm = 0.0;  // :1509
    tsum = tsum + dtm;
    if (col > 0 && dtm != 0.0)
      for (int j = 0; j < col2; ++j) c[j] = c[j] + dtm * p[j];  // :1526
    last_tsum = tsum;
    iter_seen++;
    if (col > 0) {
      // p = W'd over the variables that still move = the free variables: with it W'Z r needs no
      // pass over W (subspace_closed_form).  Not when p is what little is left of a much larger p
      // (nor after a walk of more than 2^20 segments: the host corrections of formk's new row
      // are then no longer small change).
      double pm = 0.0;
      for (int j = 0; j < col2; ++j) p_fin[j] = p[j], pm = std::max(pm, std::fabs(p[j]));
      closed_ok = nseg <= (1 << 20) && pm >= 1e-3 * p_ini_max && p_ini_max > 0.0;
    }
    return leave(tsum, last_t, last_i);
  }

