// solver_walk.inl -- member functions of Solver<T> (included inside the class body in solver.hip): the
// generalized Cauchy point itself (reference src/lbfgsb.f90:1157-1532) -- its functional form (tsum +
// iwhere), the rows the walk fixes, and cauchy(): the exact host replay of the breakpoint walk
// (:1378-1497) over the records the provider (solver_provider.inl) delivers in order.
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
    double lt2;  // the breakpoint crossed before the last one (-1: none)
  };
  static __attribute__((noinline)) int walk_raw_col0(const double *raw, size_t &pos_io, size_t end, double theta,
                                                     double clampv, bool all_n, bool bnded, WalkRaw &w,
                                                     bool &tie) {
    const double INFL = 1.0 + 16.0 * std::numeric_limits<double>::epsilon();
    const double inf = std::numeric_limits<double>::infinity();
    double f1_ = w.f1, f2_ = w.f2, dtm_ = w.dtm, tsum_ = w.tsum, tj_ = w.tj, lt_ = w.lt, lt2_ = w.lt2;
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
      lt2_ = lt_;
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
    w = WalkRaw{f1_, f2_, dtm_, tsum_, tj_, lt_, nleft_, lt2_};
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
    nrc_clear();
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
      if (debug_walk) std::fprintf(stderr, "[cauchy] own scan pass (col %d, theta %g)\n", col, theta);
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
    // (hpsolb :2079); the two differ in effect only if SOME order of a group's members makes the walk
    // end inside the group: the derivative f1 at the group's breakpoint, plus the members' jumps taken in
    // that order, turns positive before the last member (then dtm < 0 = dt, :1416).  The jumps
    // dibp^2 - theta dibp zibp + dibp w'Mc do not depend on the order (c stands still while dt = 0), so a
    // group can do that only if f1 on arrival + the sum of its POSITIVE jumps is positive.  With B = theta I
    // every jump is positive: only the group the walk ends in or right behind.  That is detected (tie_split:
    // the walk stops at a breakpoint equal to the last one it crossed; grp_sens: a crossed group of two or
    // more could have stopped it), counted, and the walk is then replayed from its start in the reference's
    // own order (exact_init / refill_exact) -- unless LBFGSB_F_INDEX_TIES opts out.
    const bool can_exact = !(flags & LBFGSB_F_INDEX_TIES);
    // (a replay would print the walk twice: under iprint >= 99 the walk runs in that order from the
    //  start; option "exact_always": every walk in that order, for tests)
    bool exact_run = can_exact && (print_level >= 99 || exact_always);
    std::vector<double> p_start(p, p + col2);
    const double f1_start = f1, f2_start = f2, dtm_start = dtm;
    for (;;) {  // at most two trips: the second one in exact order
    bool tie_split = false;
    bool grp_sens = false;
    double t_prev = -1.0;  // the breakpoint crossed before the last one
    double grp_t = -1.0, grp_f1_in = 0.0, grp_pos = 0.0;  // (col > 0) the group being crossed
    int64_t grp_n = 0;
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
            WalkRaw w{f1, f2, dtm, tsum, tj, last_t, nleft, t_prev};
            const size_t pos0 = pos;
            bool tie_ = false;
            const int code = walk_raw_col0(pv.raw, pos, end, theta, epsmch * f2_org, nbreak == nglob, bnded, w, tie_);
            const double f1_ = w.f1, f2_ = w.f2, dtm_ = w.dtm, tsum_ = w.tsum, tj_ = w.tj, lt_ = w.lt;
            const int64_t nleft_ = w.nleft;
            const double *const raw = pv.raw;
            const int64_t took = (int64_t)(pos - pos0);
            f1 = f1_, f2 = f2_, dtm = dtm_, tsum = tsum_, tj = tj_, last_t = lt_, nleft = nleft_, t_prev = w.lt2;
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
            t_prev = last_t;
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
          if (std::isfinite(hi_need)) {
            const double base = last_t > 0 ? last_t : 0.0;
            // (no slack behind a long walk: a quarter more of 10^4 ... 10^5 records to sort, gather and carry over)
            const double first = last_walk_nseg <= 4096 ? 1.0 + win_slack : 1.0;
            hi = base + (hi_need - base) * (pv.grow > 0 ? std::ldexp(1.0, std::min(pv.grow, 40)) : first);
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
        t_prev = last_t;
        last_t = tj;
        last_i = rec_gi;
        if (pv.exact || fixlist.size() < FIX_CAP)
          fixlist.push_back(rec_gi * 2 + (dibp > 0.0 ? 1 : 0));
        else
          fix_overflow = true;
        if (col > 0 && nr_flag(col)) {
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
        const double f1_arrival = f1 + dt * f2;  // (the derivative at this breakpoint, nothing fixed yet)
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
        {  // the group of equal breakpoints this one belongs to: could another order have ended the walk in it?
          const double jump = f1 - f1_arrival;
          if (grp_n > 0 && tj == grp_t) {
            grp_n++, grp_pos += std::max(jump, 0.0);
          } else {
            grp_t = tj, grp_n = 1, grp_f1_in = f1_arrival, grp_pos = std::max(jump, 0.0);
          }
          if (grp_n >= 2 && grp_f1_in + grp_pos > -1e-12 * (std::fabs(grp_f1_in) + grp_pos)) grp_sens = true;
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
    // (no pair stored: the jumps are positive, only the last group counts -- two or more crossed at last_t and
    //  the derivative non-negative behind them: dtm <= 0, or the 0 the all-breakpoints-crossed exit leaves)
    if (col == 0 && last_t >= 0.0 && t_prev == last_t && dtm <= 1e-9 * last_t) grp_sens = true;
    if ((tie_split || grp_sens) && !exact_run) {
      ntiesplit++;
      if (can_exact) {  // replay from the start of the walk, in the reference's order
        exact_run = true;
        std::copy(p_start.begin(), p_start.end(), p);
        for (int j = 0; j < col2; ++j) c[j] = 0.0;
        f1 = f1_start, f2 = f2_start, dtm = dtm_start, tsum = 0.0, nseg = 1;
        last_t = -1.0, last_i = -1;
        fixlist.clear();
        fix_overflow = false;
        nrc_clear();
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
    last_walk_nseg = nseg;
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
