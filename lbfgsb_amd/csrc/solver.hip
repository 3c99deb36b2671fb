// solver.hip -- host side of the MI355X-native L-BFGS-B inner iteration.
//
// Mirrors the reverse-communication state machine of the reference `mainlb`
// (src/lbfgsb.f90:312-949): same task protocol, same isave/dsave/lsave slots,
// same failure/refresh branches.  Every n-dimensional operation is a kernel
// launch from the k_*.hip files; every 2m x 2m operation is host code from
// host_dense.hpp.  One host thread per context, one HIP stream per context.
//
// One steady-state iteration on a bounded problem with col <= 32 pairs stored (DESIGN.md 4a
// has the why) -- two passes over W (beyond 21 pairs the first one is two launches over half of
// the columns each, k_update.hip "the split pass"):
//   FG_LNSRCH entry : update_scan as the evaluation of the first trial point  [read-only, 1 sync]
//                     (g'd and |proj g| for dcsrch; if accepted also matupd's sums, the next
//                      cauchy scan's sums and formk's new row) -- later trials: lnsrlb_eval
//   NEW_X entry     : host matupd/formt from those sums (the new pair stays pending)
//     cauchy        : host walk; window/gather only when it passes the first breakpoint
//     freev         : freev_count                                             [1 sync]
//     formk         : patches for rows that changed status (sparse), host assembly + 2 Cholesky
//     subsm+lnsrlb  : W'Z r in closed form on the host (subspace_closed_form); subsm_update:
//                     Newton step, projection, line-search set-up, first trial x, commits the
//                     pending pair                                            [stores, 1 sync]
// Fallback with a third pass (cmprlb_wtv: r, W'r, formk's new row): few free variables, long walks,
// option two_pass_maxcol.  Other paths (col = 0, restarts, unconstrained): projgr, cauchy_scan,
// cauchy_finish, formk_gram, update_pairs, lnsrlb_begin/step, pair_commit, xcp_fill,
// subsm_dir/backtrack.
//
// Source layout: this file holds the context (allocation, the reductions across ranks) and the
// mainlb state machine itself -- nine phase functions over one Mainlb struct, driven by drive(); the
// member functions of the three big phases live in files of their own, included inside the class:
//   solver_provider.inl  the Cauchy point: breakpoint provider (windows, sorts, gathers, merges over ranks)
//   solver_walk.inl      ... its functional form, the exact host walk: cauchy()
//   solver_pgcp.inl      ... the opt-in parallel search
//   solver_subspace.inl  formk, cmprlb, subsm (the closed form, the storing pass, backtracking)
//   solver_wide.inl      m > 32: the same steps out of unfused tile primitives (k_wide.hip)
//   solver_state.inl     export / import in the reference's wa / iwa layout, per-kernel doors
//   solver_doors.inl     routine doors: one routine of the reference each, on the state of the context
// Two device-pointer entries share the state machine: setulb_dev (the caller's x and g in place,
// t = x / r = g as copies) and setulb_dev_pp (two caller buffer pairs that swap roles, no copies).
//
// There is no CPU fallback anywhere in these files.
#include "solver_base.hpp"

namespace {

template <typename T>
class Solver final : public lbfgsb_hip_ctx {
 public:
  // ---- device state ----
  T *ws = nullptr, *wy = nullptr, *zero_buf = nullptr;
  int64_t ld = 0;
  T *z = nullptr, *d = nullptr, *xp = nullptr, *tbrk = nullptr;
  // t (the iterate at the start of the current line search, lnsrlb :2235) and r (its gradient,
  // :2236) are ROLES: the kernels read them through these two pointers.  With the classic entry
  // they always point at the context's own buffers (t_own, r_own), which the line-search set-up
  // fills with copies of x and g.  With ping-pong iterate buffers (setulb_dev_pp) the set-up copies
  // nothing: x and g stay where they are and BECOME t and r, the trial point goes to the other pair.
  T *t = nullptr, *r = nullptr, *t_own = nullptr, *r_own = nullptr;
  bool pp = false;        // this run uses the ping-pong entry
  int entry_mode = 0;     // which entry this run belongs to: 0 not decided (fresh context, or a state was
                          // imported: the next call decides), 1 setulb_dev, 2 setulb_dev_pp
  T *xb[2] = {nullptr, nullptr}, *gb[2] = {nullptr, nullptr};
  int pp_cur = 0;         // the pair the last return referred to
  const T *x_lean = nullptr;  // where the first trial point of a lean subspace pass lives (d = x_lean - t)
  // ---- uniform bounds: l, u, nbd that hold ONE value each (the box [a, b]^n, x >= 0, ...) are not
  //      streamed by the passes over W: the kernels read a 64-byte constant buffer instead (8 + 8 + 1
  //      bytes per row less in each of the two passes; kernels.hpp `ub`).  Detected at START by
  //      errclb's pass, bit for bit; l, u, nbd must not change during a run anyway. ----
  //      FEW-VALUED bound arrays (<= 8 distinct values each: driver3's alternating box, test/driver3.f90:102-120)
  //      are dictionary-coded (bit 3, lbk::UB_DICT): the one-byte nbd copy carries nbd | l-index << 2 |
  //      u-index << 5 and the two buffers hold the value tables (kernels_common.hpp).
  //      The caller's arrays are compared with this snapshot every bcheck_every iterations (bounds_verify). ----
  bool ub_on = true;          // (option "uniform_bounds")
  bool dict_on = true;        // (option "dict_bounds")
  int bcheck_every = 32;      // (option "bounds_check": 0 = never)
  int64_t nbounds_checks = 0;
  lbk::BoundTables ub_tab{};  // the values the passes use where an array is not streamed
  int ub_mask = 0;            // bit 0 l, bit 1 u, bit 2 nbd, bit 3 dictionary (with bits 0, 1)
  char *ub_buf = nullptr;     // 3 x 64 bytes on the device
  const void *ub_l = nullptr, *ub_u = nullptr;  // the caller arrays the detection looked at
  const int32_t *ub_nbd = nullptr;
  const T *lk(const T *l) const { return (ub_mask & 1) ? (const T *)ub_buf : l; }
  const T *uk(const T *u) const { return (ub_mask & 2) ? (const T *)(ub_buf + 64) : u; }
  const lbk::nb_t *nbk() const { return (ub_mask & 4) ? (const lbk::nb_t *)(ub_buf + 128) : nbd8; }
  lbk::iw_t *iwhere = nullptr;  // one byte per row (the reference's int32 only in export/import)
  int32_t *index = nullptr, *indx2 = nullptr, *scan_tmp = nullptr;
  int8_t *wasfree = nullptr, *prevfree = nullptr;
  // cauchy selection
  static constexpr uint32_t SEL_CAP = 1u << 18;
  static constexpr uint32_t CHUNK_MAX = 16384;
  uint64_t *keys[2] = {nullptr, nullptr};
  uint32_t *idx[2] = {nullptr, nullptr};
  size_t sel_alloc = 0;  // elements allocated in keys/idx
  void *sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  uint32_t *d_count = nullptr, *h_count = nullptr;
  uint32_t *d_fcount = nullptr;  // freev's two list-position counters, used in turn (k_freev.hip)
  int fv_parity = 0;
  std::vector<double> h_loc;     // this rank's OWN values of the last fetch (before the reduction over ranks)
  // rows whose free/active status changed in the last freev (formk patches)
  static constexpr uint32_t CHG_CAP = 1u << 18;
  uint32_t *d_chg = nullptr;
  uint32_t chg_local = 0;
  double *d_msg = nullptr, *d_msg_all = nullptr, *h_msg_all = nullptr, *h_msg_loc = nullptr;
  // single rank: the next chunk of walk records is gathered and copied while the host walks the
  // current one (second pair of message buffers; h_msg_loc doubles as the landing buffer)
  double *d_msg2 = nullptr;
  hipEvent_t pf_ev = nullptr;
  bool pf_valid = false;
  uint32_t pf_pl = 0, pf_len = 0, pf_rem = 0;
  int pf_cur = 0;
  double *h_hdr = nullptr;
  size_t msg_len = 0;  // doubles per rank message
  // several ranks with a communicator: the all-gathered chunks are merged on the device (one sort of
  // <= nranks * chunk keys) and arrive on the host as ONE ordered run (solver_provider.inl, refill)
  uint64_t *mg_keys[2] = {nullptr, nullptr};
  uint32_t *mg_vals[2] = {nullptr, nullptr};
  void *mg_tmp = nullptr;
  size_t mg_tmp_bytes = 0, mg_slots = 0;
  double *d_merged = nullptr;
  // doubles per rank of the merged message: records + 4 header doubles + one byte per record
  size_t mg_stride() const { return msg_len + msg_len / 32 + 16; }
  // reductions (several ranks: every rank's partials, rank-major, see fetch)
  double *h_res = nullptr, *d_res_all = nullptr, *h_res_all = nullptr;
  size_t res_len = 0;
  // streams
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // host matrices (reference layouts, column-major)
  std::vector<double> sy, ss, wt, wn, snd, wa8m;
  // comm
  ncclComm_t comm = nullptr;
  lbfgsb_allreduce_fn cb_ar = nullptr;
  lbfgsb_allgather_fn cb_ag = nullptr;
  void *cb_user = nullptr;
  // report
  lbr::Report rep;
  char word[4] = {'-', '-', '-', 0};
  int64_t err_k = 0;
  bool quiet = false;  // ranks > 0 never print
  const bool debug_walk = std::getenv("LBFGSB_DEBUG") != nullptr;  // trace of the breakpoint walk

  ~Solver() override { release(); }

  void release() {
    if (debug_walk && n_mid > 0)
      std::fprintf(stderr, "[host] %lld stretches, us each: linesearch+return %.1f | caller %.1f | update %.1f | "
                           "cauchy+freev %.1f | formk+subsm algebra %.1f | all %.1f\n",
                   (long long)n_mid, t_seg[0] / n_mid * 1e6, t_seg[1] / n_mid * 1e6, t_seg[2] / n_mid * 1e6,
                   t_seg[3] / n_mid * 1e6, t_seg[4] / n_mid * 1e6, t_mid / n_mid * 1e6);
    auto F = [](auto *&p) {
      if (p) (void)hipFree(p);
      p = nullptr;
    };
    F(ws), F(wy), F(lmask), F(zero_buf), F(z), F(r_own), F(d), F(t_own), F(xp), F(tbrk), F(iwhere), F(nbd8), F(index), F(indx2),
        F(scan_tmp), F(wasfree), F(prevfree), F(keys[0]), F(keys[1]), F(idx[0]), F(idx[1]),
        F(sort_tmp), F(d_count), F(d_chg), F(d_msg), F(d_msg2), F(d_msg_all), F(q.d_part), F(q.d_res), F(q.d_gpart),
        F(q.d_part_alt[0]), F(q.d_part_alt[1]), F(q.d_fin_count), F(d_fcount), F(wide_rows),
        F(d_fix), F(pg_buf), F(pg_tmp), F(sp_keys), F(sp_idx), F(sp_count), F(sp_msg), F(sp_msg_all), F(d_res_all),
        F(ub_buf), F(mg_keys[0]), F(mg_keys[1]), F(mg_vals[0]), F(mg_vals[1]), F(mg_tmp), F(d_merged);
    F(hx), F(hg), F(hl), F(hu), F(hnbd);
    auto H = [](auto *&p) {
      if (p) (void)hipHostFree(p);
      p = nullptr;
    };
    H(h_count), H(h_msg_all), H(h_msg_loc), H(h_hdr), H(h_res), H(h_flag), H(h_pub), H(h_fin_flag), H(h_fix), H(h_sp_all), H(h_sp_loc), H(h_res_all);
    if (pf_ev) (void)hipEventDestroy(pf_ev);
    pf_ev = nullptr;
    if (order_ev) (void)hipEventDestroy(order_ev);
    order_ev = nullptr;
    if (return_ev) (void)hipEventDestroy(return_ev);
    return_ev = nullptr;
    for (auto &ring : clk_ev)
      for (auto &pair : ring)
        for (auto &e : pair) {
          if (e) (void)hipEventDestroy(e);
          e = nullptr;
        }
    if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
    comm = nullptr;
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
    if (rep.itf) std::fclose(rep.itf);
    rep.itf = nullptr;
  }

  int init(int64_t n_, int64_t nglob_, int64_t row0_, int m_, int flags_, int device_,
           void *stream_) {
    n = n_, nglob = nglob_, row0 = row0_, m = m_, flags = flags_, device = device_;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(LBFGSB_E_NOGPU, "no HIP device visible (this library has no CPU path)");
    HIPCHK(hipSetDevice(device));
    if (stream_) {
      stream = (hipStream_t)stream_;
    } else {
      HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      own_stream = true;
    }
    q.stream = stream;
    defer_on = (flags & LBFGSB_F_DEFER_LNSRCH) != 0;
    ld = ((n + lbk::CW_TILE - 1) / lbk::CW_TILE) * lbk::CW_TILE;  // (whole layout tiles)
    // streamed-once data: nontemporal loads unless W fits the 256 MiB Infinity Cache
    q.nt = (size_t)2 * ld * m * sizeof(T) > ((size_t)192 << 20);
    const size_t wbytes = (size_t)ld * m * sizeof(T);
    HIPCHK(hipMalloc(&ws, wbytes));
    HIPCHK(hipMalloc(&wy, wbytes));
    HIPCHK(hipMemsetAsync(ws, 0, wbytes, stream));
    HIPCHK(hipMemsetAsync(wy, 0, wbytes, stream));
    {
      const size_t nw = (size_t)((n + lbk::CW_TILE - 1) / lbk::CW_TILE) * (lbk::CW_TILE / 64) + 4;
      HIPCHK(hipMalloc(&lmask, nw * sizeof(uint64_t)));
      HIPCHK(hipMemsetAsync(lmask, 0, nw * sizeof(uint64_t), stream));
      lbk::launch_lmask_ones(q, n, lmask);
    }
    HIPCHK(hipMalloc(&ub_buf, 192));
    // read by the unroll slots beyond the stored pairs (16 bytes per lane from its start; under the tile-local
    // layout of W at a lane's in-tile slot: a tile of the widest kind long)
    HIPCHK(hipMalloc(&zero_buf, 2048));
    HIPCHK(hipMemsetAsync(zero_buf, 0, 2048, stream));
    const size_t vb = (size_t)(n + 32) * sizeof(T);
    for (T **p : {&z, &r_own, &d, &t_own, &xp, &tbrk}) {
      HIPCHK(hipMalloc(p, vb));
      HIPCHK(hipMemsetAsync(*p, 0, vb, stream));
    }
    t = t_own, r = r_own;
    HIPCHK(hipMalloc(&iwhere, (size_t)(n + 32) * sizeof(lbk::iw_t)));
    HIPCHK(hipMalloc(&nbd8, (size_t)(n + 32) * sizeof(lbk::nb_t)));
    HIPCHK(hipMemsetAsync(iwhere, 0, (size_t)(n + 32) * sizeof(lbk::iw_t), stream));
    HIPCHK(hipMalloc(&wasfree, (size_t)n + 32));
    HIPCHK(hipMemsetAsync(wasfree, 1, (size_t)n + 32, stream));
    if (flags & LBFGSB_F_MIRROR_INDEX) {
      HIPCHK(hipMalloc(&prevfree, (size_t)n + 32));
      HIPCHK(hipMemsetAsync(prevfree, 1, (size_t)n + 32, stream));
      HIPCHK(hipMalloc(&index, (size_t)n * sizeof(int32_t)));
      HIPCHK(hipMalloc(&indx2, (size_t)n * sizeof(int32_t)));
      HIPCHK(hipMemsetAsync(index, 0, (size_t)n * sizeof(int32_t), stream));
      HIPCHK(hipMemsetAsync(indx2, 0, (size_t)n * sizeof(int32_t), stream));
      const size_t nch = (size_t)((n + 1023) / 1024) + 2;
      HIPCHK(hipMalloc(&scan_tmp, 3 * nch * sizeof(int32_t)));
    }
    // reduction scratch
    const size_t E = (size_t)2 * m * m + m;
    if (m > lbk::MAXM) {
      DEFER_OFF = 8 * lbk::maxc_stride(m - 1) + 24, SPEC_OFF = DEFER_OFF + 8;
      q.split_base_min = DEFER_OFF + 8;  // (<= split_base(m, 1), which split_res_len sizes d_res for)
    }
    // (the widest phase or a from-scratch Gram; behind DEFER_OFF the four deferred sums, behind SPEC_OFF the
    //  speculative freev counts + formk patch of a trial point: 4 + E + 1)
    res_len = std::max<size_t>(std::max<size_t>(lbk::RES_MAX, E) + 8, (size_t)SPEC_OFF + 4 + E + 1 + 8);
    res_len = std::max<size_t>(res_len, lbk::split_res_len(m));  // (the parts of a split update pass)
    HIPCHK(hipMalloc(&q.d_part, (size_t)lbk::RES_MAX * lbk::MAX_BLOCKS * sizeof(double)));
    HIPCHK(hipMalloc(&q.d_res, res_len * sizeof(double)));
    HIPCHK(hipMalloc(&q.d_gpart, (E + 1) * lbk::GRAM_BLOCKS * sizeof(double)));  // (+ the eager patch's flag slot)
    HIPCHK(hipHostMalloc(&h_res, res_len * sizeof(double)));
    HIPCHK(hipHostGetDevicePointer((void **)&hd_res, h_res, 0));
    HIPCHK(hipHostMalloc(&h_flag, 64));
    std::memset(h_flag, 0, 64);
    HIPCHK(hipHostGetDevicePointer((void **)&hd_flag, h_flag, 0));
    // finalize as publisher: the host mirror of d_res, its sequence word, the workgroup counter; two small
    // partial-sum matrices for kernels whose finalize is parked (kernels.hpp, Queue)
    HIPCHK(hipHostMalloc(&h_pub, res_len * sizeof(double)));
    HIPCHK(hipHostGetDevicePointer((void **)&q.hd_pub, h_pub, 0));
    HIPCHK(hipHostMalloc(&h_fin_flag, 64));
    std::memset(h_fin_flag, 0, 64);
    HIPCHK(hipHostGetDevicePointer((void **)&q.hd_fin_flag, h_fin_flag, 0));
    HIPCHK(hipMalloc(&q.d_fin_count, 64));
    HIPCHK(hipMemsetAsync(q.d_fin_count, 0, 64, stream));
    for (double *&pa : q.d_part_alt)
      HIPCHK(hipMalloc(&pa, (size_t)lbk::Queue::ALT_SLOTS * lbk::MAX_BLOCKS * sizeof(double)));
    q.fin_publish = spin_on;
    // cauchy selection scratch (window mode); the full-sort buffers grow on demand
    CHK(ensure_sel(SEL_CAP));
    HIPCHK(hipMalloc(&d_count, sizeof(uint32_t)));
    HIPCHK(hipMalloc(&d_fcount, 64));
    HIPCHK(hipMemsetAsync(d_fcount, 0, 64, stream));
    h_loc.assign(res_len, 0.0);
    HIPCHK(hipMalloc(&d_chg, (size_t)CHG_CAP * sizeof(uint32_t)));
    HIPCHK(hipHostMalloc(&h_count, sizeof(uint32_t)));
    msg_len = 2 + (size_t)CHUNK_MAX * (2 * m + 4);
    HIPCHK(hipMalloc(&d_msg, msg_len * sizeof(double)));
    HIPCHK(hipMalloc(&d_msg2, msg_len * sizeof(double)));
    HIPCHK(hipEventCreateWithFlags(&pf_ev, hipEventDisableTiming));
    HIPCHK(hipHostMalloc(&h_hdr, 2 * sizeof(double)));
    HIPCHK(hipMalloc(&sp_keys, SPEC_CAP * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&sp_idx, SPEC_CAP * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&sp_count, sizeof(uint32_t)));
    HIPCHK(hipMalloc(&sp_msg, (2 + (size_t)SPEC_CAP * (2 * m_ + 4)) * sizeof(double)));
    HIPCHK(hipMalloc(&d_fix, FIX_CAP * sizeof(int64_t)));
    HIPCHK(hipHostMalloc(&h_fix, FIX_CAP * sizeof(int64_t)));
    CHK(set_ranks(0, 1));
    sy.assign((size_t)m * m, 0.0);
    ss.assign((size_t)m * m, 0.0);
    wt.assign((size_t)m * m, 0.0);
    wn.assign((size_t)4 * m * m, 0.0);
    snd.assign((size_t)4 * m * m, 0.0);
    wa8m.assign((size_t)8 * m, 0.0);
    scan.p.assign((size_t)2 * std::max(m, lbk::MAXM), 0.0);
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  // (re)size the all-gather landing buffers for `nr` ranks
  int set_ranks(int rk, int nr) {
    rank = rk, nranks = nr;
    if (d_msg_all) (void)hipFree(d_msg_all);
    if (h_msg_all) (void)hipHostFree(h_msg_all);
    if (h_msg_loc) (void)hipHostFree(h_msg_loc);
    d_msg_all = h_msg_all = h_msg_loc = nullptr;
    HIPCHK(hipMalloc(&d_msg_all, (size_t)nr * msg_len * sizeof(double)));
    HIPCHK(hipHostMalloc(&h_msg_all, (size_t)nr * mg_stride() * sizeof(double)));
    for (auto **p : {(void **)&mg_keys[0], (void **)&mg_keys[1], (void **)&mg_vals[0], (void **)&mg_vals[1],
                     (void **)&mg_tmp, (void **)&d_merged}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    mg_slots = 0;
    HIPCHK(hipHostMalloc(&h_msg_loc, msg_len * sizeof(double)));
    if (sp_msg_all) (void)hipFree(sp_msg_all);
    if (h_sp_all) (void)hipHostFree(h_sp_all);
    if (h_sp_loc) (void)hipHostFree(h_sp_loc);
    sp_msg_all = h_sp_all = h_sp_loc = nullptr;
    HIPCHK(hipMalloc(&sp_msg_all, (size_t)nr * sp_len() * sizeof(double)));
    HIPCHK(hipHostMalloc(&h_sp_all, ((size_t)nr * sp_len() + nr) * sizeof(double)));
    HIPCHK(hipHostMalloc(&h_sp_loc, sp_len() * sizeof(double)));
    spcand.valid = false;
    if (d_res_all) (void)hipFree(d_res_all);
    if (h_res_all) (void)hipHostFree(h_res_all);
    d_res_all = h_res_all = nullptr;
    HIPCHK(hipMalloc(&d_res_all, (size_t)nr * res_len * sizeof(double)));
    HIPCHK(hipHostMalloc(&h_res_all, (size_t)nr * res_len * sizeof(double)));
    HIPCHK(hipHostGetDevicePointer((void **)&hd_res_all, h_res_all, 0));
    return 0;
  }

  int ensure_sel(size_t count) {
    if (count <= sel_alloc) return 0;
    auto F = [](auto *&p) {
      if (p) (void)hipFree(p);
      p = nullptr;
    };
    F(keys[0]), F(keys[1]), F(idx[0]), F(idx[1]), F(sort_tmp);
    sel_alloc = 0;
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipMalloc(&keys[k], count * sizeof(uint64_t)));
      HIPCHK(hipMalloc(&idx[k], count * sizeof(uint32_t)));
    }
    sort_tmp_bytes = lbk::sort_pairs_temp_bytes(count) + 256;
    HIPCHK(hipMalloc(&sort_tmp, sort_tmp_bytes));
    sel_alloc = count;
    return 0;
  }

  // ---- complete a reduction across ranks and bring it to the host ----
  // ONE collective per host sync: the k partials of every rank are all-gathered (ncclAllGather on
  // the solver's stream, k <= 8m + 15 doubles per rank) and reduced on the host in rank order --
  // sums | minima | maxima in one go, every rank gets bit-identical results by construction,
  // whatever algorithm RCCL picks for the message.  (r02: up to three grouped ncclAllReduce.)
  // How the numbers reach the host: publish_kernel writes them into mapped host memory and then a
  // sequence word, which this thread polls (option "spin" = 0: a D2H copy + hipStreamSynchronize as
  // in round 3).  The poll is bounded: after SPIN_LIMIT_S the runtime's own wait takes over, so a
  // faulted or hung stream is reported the way it always was.
  // DEFER_OFF: four more slots (3 sums + 1 minimum) behind the widest phase -- the line-search sums of a
  // storing pass that did not wait for them (LBFGSB_F_DEFER_LNSRCH); while defer_live they travel with
  // every fetch and are reduced like the rest.
  // (m > 32: the merged layout of the split update pass is 8 maxc_stride(m - 1) + 15 doubles long, its parts start 32
  //  doubles behind it (split_base): the deferred sums sit in between)
  int DEFER_OFF = lbk::RES_MAX;
  // SPEC_OFF: the sums of a SPECULATIVE freev + formk-patch chain queued behind the evaluation of a trial point
  // (phase_entry): spec_live_len slots, all sums, fetched and reduced with that evaluation's fetch
  int SPEC_OFF = lbk::RES_MAX + 8;
  int spec_live_len = 0;
  static constexpr double SPIN_LIMIT_S = 0.05;
  bool defer_live = false;
  bool spin_on = true;  // (option "spin")
  unsigned long long pub_seq = 0;
  unsigned long long *h_flag = nullptr, *hd_flag = nullptr;  // host / device view of the sequence word
  // ... of the finalize kernels that publish by themselves (single rank: k_misc.hip, finalize_kernel), and
  // the host mirror of d_res they write; fin_seq_seen = the last launch a fetch has waited for
  unsigned long long *h_fin_flag = nullptr, fin_seq_seen = 0;
  bool tail_copy_queued = false;  // a D2H copy the next fetch has to cover was queued behind the last finalize
  double *h_pub = nullptr;
  double *hd_res = nullptr, *hd_res_all = nullptr;           // device views of h_res / h_res_all
  int wait_published(unsigned long long seq, const unsigned long long *flag = nullptr) {
    if (!flag) flag = h_flag;
    const double t0 = now_s();
    for (unsigned it = 1;; ++it) {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if ((it & 0x3ff) == 0 && now_s() - t0 > SPIN_LIMIT_S) {
        HIPCHK(hipStreamSynchronize(stream));
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
          return fail(LBFGSB_E_NOGPU, "the stream finished without publishing its results");
        break;
      }
    }
    t_wait += now_s() - t0;
    return 0;
  }
  int fetch(int nsum, int nmin, int nmax) {
    const int k = nsum + nmin + nmax;
    if (q.launch_err != hipSuccess) {  // a kernel launch of this phase failed: name it
      const hipError_t e = q.launch_err;
      q.launch_err = hipSuccess;
      return fail(LBFGSB_E_NOGPU, std::string("kernel launch failed in ") +
                                      (q.launch_err_where ? q.launch_err_where : "?") + ": " +
                                      hipGetErrorString(e));
    }
    if (defer_live && k > DEFER_OFF) return fail(LBFGSB_E_STATE, "fetch: phase overlaps the deferred line-search sums");
    if (spec_live_len && k > DEFER_OFF) return fail(LBFGSB_E_STATE, "fetch: phase overlaps the speculative sums");
    const int kk = spec_live_len ? SPEC_OFF + spec_live_len : (defer_live ? DEFER_OFF + 4 : k);  // doubles that travel
    if ((size_t)kk > res_len) return fail(LBFGSB_E_STATE, "fetch: more partials than the buffer holds");
    if (comm || nranks > 1) ncoll++, coll_bytes += (int64_t)kk * 8;
    const double *src = q.d_res;
    double *dst = h_res, *dst_dev = hd_res;
    size_t cnt = (size_t)kk;
    if (comm) {
      if (g_rccl.AllGather(q.d_res, d_res_all, (size_t)kk, ncclDouble, comm, stream) != ncclSuccess)
        return fail(LBFGSB_E_COMM, "ncclAllGather of the partial sums failed");
      src = d_res_all, dst = h_res_all, dst_dev = hd_res_all, cnt = (size_t)nranks * kk;
    }
    lbk::finalize_flush(q);  // (parked reductions that no later kernel has taken along)
    // (only if that finalize launch IS the tail of the stream: a kernel or a copy queued behind it -- the
    //  candidate records of spec_queue -- would not be covered by its sequence word)
    const bool fin_is_tail = q.launches == q.launches_at_fin && !tail_copy_queued;
    tail_copy_queued = false;
    if (spin_on && !comm && q.fin_publish && q.fin_seq != fin_seq_seen && fin_is_tail) {
      // single rank: the finalize kernels of this phase have mirrored their results into host memory
      // themselves; the last one stored its sequence number when all of it was out
      CHK(wait_published(q.fin_seq, h_fin_flag));
      fin_seq_seen = q.fin_seq;
      std::memcpy(h_res, h_pub, cnt * sizeof(double));
    } else if (spin_on) {
      lbk::launch_publish(q, src, dst_dev, (int)cnt, ++pub_seq, hd_flag);
      CHK(wait_published(pub_seq));
    } else {
      HIPCHK(hipMemcpyAsync(dst, src, cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
      const double t0 = now_s();
      HIPCHK(hipStreamSynchronize(stream));
      t_wait += now_s() - t0;
    }
    nsync++;
    if (clock_on) clk_collect();
    std::memcpy(h_loc.data(), comm ? h_res_all + (size_t)rank * kk : h_res, (size_t)kk * sizeof(double));
    if (comm) {
      auto over_ranks = [&](int j, int op) {  // 0 sum, 1 min, 2 max -- in rank order
        double v = h_res_all[j];
        for (int rk = 1; rk < nranks; ++rk) {
          const double w = h_res_all[(size_t)rk * kk + j];
          v = op == 0 ? v + w : (op == 1 ? std::fmin(v, w) : std::fmax(v, w));
        }
        h_res[j] = v;
      };
      for (int j = 0; j < k; ++j) over_ranks(j, j < nsum ? 0 : (j < nsum + nmin ? 1 : 2));
      if (defer_live)
        for (int j = 0; j < 4; ++j) over_ranks(DEFER_OFF + j, j < 3 ? 0 : 1);
      for (int j = 0; j < spec_live_len; ++j) over_ranks(SPEC_OFF + j, 0);
    } else if (nranks > 1) {
      if (!cb_ar) return fail(LBFGSB_E_COMM, "multi-rank context without a reducer");
      if (cb_ar(cb_user, h_res, nsum, nmin, nmax) != 0)
        return fail(LBFGSB_E_COMM, "host all-reduce callback failed");
      if (defer_live && cb_ar(cb_user, h_res + DEFER_OFF, 3, 1, 0) != 0)
        return fail(LBFGSB_E_COMM, "host all-reduce callback failed");
      if (spec_live_len && cb_ar(cb_user, h_res + SPEC_OFF, spec_live_len, 0, 0) != 0)
        return fail(LBFGSB_E_COMM, "host all-reduce callback failed");
    }
    return 0;
  }

  // ---- tile-local free-row layout of W ("compact W": k_layout.hip; option "compact_w") ----
  uint64_t *lmask = nullptr;   // one bit per row (ceil128(n) bits): set = the row sits in the front run of its tile
  // option "compact_w": 0 off; 1 the two passes over W run on the layout WHILE IT IS PACKED (natural-order kernels
  // before the first pack and whenever something asked for natural order: a problem whose rows are all free never
  // pays for the option); 2 they always do (the sums then do not depend on when the layout is made: the tests)
  int cw_mode = 0;
  bool cw_on = false;          // cw_mode != 0 (fp64, m <= 10, no mirroring: cw_eligible)
  int64_t cw_min_rows = -1;    // option "compact_min_rows": automatic policy packs only from this many rows on
                               // (-1: when W is much larger than the Infinity Cache, as the nontemporal loads)
  bool cw_packed = false;      // some bit is clear: the columns are NOT in natural order
  int cw_policy = 1;           // option "compact_policy": 0 never pack, 1 automatic, 2 re-pack in every iteration
  int64_t cw_stale = 0;        // rows (all ranks) that changed status since the layout was made
  int64_t ncw_pack = 0, ncw_unpack = 0;
  int cw_hold = 0;             // iterations the automatic policy waits after something asked for natural order
  int live_head = 1, live_col = 0;  // the columns of W that hold pairs (what a re-sort has to move)
  bool cw_eligible() const {
    return cw_on && sizeof(T) == 8 && m <= 10 && !(flags & LBFGSB_F_MIRROR_INDEX);
  }
  lbk::WStore<T> Wraw() const { return lbk::WStore<T>{ws, wy, ld, m, zero_buf}; }
  // natural row order -- for every kernel that does not know the layout (fallback passes, Gram, doors, export)
  lbk::WStore<T> W() {
    if (cw_packed) {
      lbk::launch_w_relayout<T>(q, n, nullptr, lmask, Wraw(), live_head, live_col);
      cw_packed = false, cw_stale = 0, ncw_unpack++;
      cw_hold = std::min(64, std::max(4, 2 * cw_hold));  // (a path that keeps asking: back off)
    }
    return Wraw();
  }
  // the layout as it is, for the kernels that take it into account
  lbk::WStore<T> Wc() {
    lbk::WStore<T> w = Wraw();
    if (cw_eligible() && (cw_mode == 2 || cw_packed)) w.lmask = lmask;
    return w;
  }
  // Re-sort the tiles so that the rows that are free NOW (iwhere <= 0, after the walk) come first.  Called in front
  // of the storing pass, where iwhere is final for the iteration.  Automatic policy (compact_policy = 1):
  //   * pack once a tenth of the rows is not free, the free set has SETTLED -- two iterations in a row that each
  //     changed < 0.5 % of the rows --, the memory is full (col = m) and W is much larger than the Infinity Cache;
  //   * from then on re-sort whenever a row has changed status since: the kernel skips the tiles whose bits stand,
  //     so its cost is a scan of iwhere (1 byte per row) plus one read + write of the live columns of the DIRTY
  //     tiles (a changed row costs 128 rows x 2 col x 16 bytes once; left alone it costs a slow fetch in every pass);
  //   * an iteration that moves more than 2 % of the rows (driver3's Rosenbrock at n = 1e7 flips 5e6 rows in and out
  //     of the free set every few iterations) ends the packing: back to natural order, and no new attempt for a while.
  // The sums do not depend on any of this under compact_w = 2 (for_tiles_cw); under compact_w = 1 the kernels of the
  // natural order run while the layout is not packed.
  int cw_settled = 0;
  void cw_maybe_pack(int head, int col, int64_t changed_now) {
    live_head = head, live_col = col;
    if (!cw_eligible() || cw_policy == 0) return;
    cw_stale += changed_now;
    bool go = cw_policy == 2;
    if (cw_policy == 1) {
      const double nn = (double)nglob;
      cw_settled = (double)changed_now <= 0.005 * nn ? cw_settled + 1 : 0;
      if (cw_packed && (double)changed_now > 0.02 * nn) {  // the free set is on the move
        (void)W();
        cw_hold = std::max(cw_hold, 8);
        return;
      }
      if (cw_hold > 0) {
        cw_hold--;
        return;
      }
      // (a W that lives in the Infinity Cache gains nothing from fewer HBM bytes: n = 1e6, m = 10 lost 2.5 %)
      const bool big = cw_min_rows >= 0 ? n >= cw_min_rows : (size_t)2 * ld * m * sizeof(T) > ((size_t)192 << 20);
      // (... and the memory is full: while it fills, the storing pass runs its general instantiation -- runtime
      //  column count, pending pair by select -- which on the layout costs 5.2 ms against 3.4 in natural order)
      if (!cw_packed)
        go = big && (double)(nglob - nfree_g) >= 0.10 * nn && cw_settled >= 2 && col >= m;
      else
        go = cw_stale > 0;
    }
    if (!go) return;
    lbk::launch_w_relayout<T>(q, n, iwhere, lmask, Wraw(), head, col);
    cw_packed = true, cw_stale = 0, ncw_pack++;
  }

#include "solver_provider.inl"  // the Cauchy point: breakpoint provider (windows, sorts, gathers, merges)
#include "solver_walk.inl"      // ... its functional form, the exact host walk: cauchy()
#include "solver_pgcp.inl"      // ... the opt-in parallel search
#include "solver_subspace.inl"  // formk, cmprlb, subsm
#include "solver_wide.inl"      // m > 32: the iteration out of unfused tile primitives
  int print_level = -1;

  // =================================================================== mainlb
  // The reference keeps mainlb's locals in lsave/isave/dsave between calls (:904-947); one call's
  // view of them, plus the call's arguments, travels through the phases below in this struct.
  struct Mainlb {
    T *x, *g;
    const T *l, *u;
    const int32_t *nbd;
    double *f;
    double factr, pgtol;
    char *task, *csave;
    int32_t *lsave, *isave;  // isave = mainlb's Isave(1:23) = the user's isave(22:44)
    double *dsave;
    int ipr;
    bool prjctd = false, cnstnd = false, boxed = false, updatd = false, wrk = false;
    int nintol = 0, iback = 0, nskip = 0, head = 0, col = 0, iter = 0, itail = 0, iupdat = 0, nseg = 0,
        nfgv = 0, info = 0, ifun = 0, iword = 0, nfree = 0, nact = 0, ileave = 0, nenter = 0;
    double theta = 0, fold = 0, tol = 0, dnorm = 0, epsmch = 0, cpu1 = 0, cachyt = 0, sbtime = 0,
           lnscht = 0, time1 = 0, gd = 0, stpmx = 0, sbgnrm = 0, stp = 0, gdold = 0, dtd = 0, xstep = 0.0;
    // this call's route through the loop, and what the entry phase found
    bool compute_pg = true, prelims = true, linesearch = true;
    double spec_sbgnrm = 0.0;
    int fo = 0;  // 1: the value of a deferred built-in objective rides in front of the first fetch
    // LBFGSB_F_DEFER_LNSRCH: this call entered with the evaluation of a trial point whose line-search
    // set-up had not happened yet (its sums arrived with this call's first fetch).  While `landing`
    // the call works on the ITERATE (x, g, f as at the set-up); the evaluation waits here
    bool landing = false, spec_holds = false;
    double f_trial = 0.0, gd_trial = 0.0, sbg_trial = 0.0, spec_iw_changed = 0.0;
  };
  // what a phase tells the driver loop: go on with the next phase | start the loop trip again
  // (memory refreshed, update skipped) | the call is over | run the same phase once more
  enum Flow { NEXT, AGAIN, DONE, REPEAT };
  static int again(Flow &fl) { return fl = AGAIN, 0; }
  static int done(Flow &fl) { return fl = DONE, 0; }
  static int repeat(Flow &fl) { return fl = REPEAT, 0; }
#define MAINLB_VIEW(L)                                                                             \
  [[maybe_unused]] T *const x = L.x;                                                              \
  [[maybe_unused]] T *const g = L.g;                                                              \
  [[maybe_unused]] const T *const l = L.l;                                                        \
  [[maybe_unused]] const T *const u = L.u;                                                        \
  [[maybe_unused]] const int32_t *const nbd = L.nbd;                                              \
  [[maybe_unused]] double *const f = L.f;                                                         \
  [[maybe_unused]] char *const task = L.task;                                                     \
  [[maybe_unused]] char *const csave = L.csave;                                                   \
  [[maybe_unused]] int32_t *const lsave = L.lsave;                                                \
  [[maybe_unused]] int32_t *const isave = L.isave;                                                \
  [[maybe_unused]] double *const dsave = L.dsave;                                                 \
  [[maybe_unused]] const double factr = L.factr, pgtol = L.pgtol;                                 \
  [[maybe_unused]] const int ipr = L.ipr;                                                         \
  [[maybe_unused]] bool &prjctd = L.prjctd, &cnstnd = L.cnstnd, &boxed = L.boxed,                 \
                        &updatd = L.updatd, &wrk = L.wrk, &compute_pg = L.compute_pg,            \
                        &prelims = L.prelims, &linesearch = L.linesearch;                         \
  [[maybe_unused]] int &nintol = L.nintol, &iback = L.iback, &nskip = L.nskip, &head = L.head,    \
                       &col = L.col, &iter = L.iter, &itail = L.itail, &iupdat = L.iupdat,        \
                       &nseg = L.nseg, &nfgv = L.nfgv, &info = L.info, &ifun = L.ifun,            \
                       &iword = L.iword, &nfree = L.nfree, &nact = L.nact, &ileave = L.ileave,    \
                       &nenter = L.nenter, &fo = L.fo;                                            \
  [[maybe_unused]] double &theta = L.theta, &fold = L.fold, &tol = L.tol, &dnorm = L.dnorm,       \
                          &epsmch = L.epsmch, &cpu1 = L.cpu1, &cachyt = L.cachyt,                 \
                          &sbtime = L.sbtime, &lnscht = L.lnscht, &time1 = L.time1, &gd = L.gd,   \
                          &stpmx = L.stpmx, &sbgnrm = L.sbgnrm, &stp = L.stp, &gdold = L.gdold,   \
                          &dtd = L.dtd, &xstep = L.xstep, &spec_sbgnrm = L.spec_sbgnrm

  void save_locals(Mainlb &L) {  // :904-947
    MAINLB_VIEW(L);
    lsave[0] = prjctd, lsave[1] = cnstnd, lsave[2] = boxed, lsave[3] = updatd;
    isave[0] = nintol, isave[2] = 0, isave[3] = iback, isave[4] = nskip, isave[5] = head;
    isave[6] = col, isave[7] = itail, isave[8] = iter, isave[9] = iupdat, isave[11] = nseg;
    isave[12] = nfgv, isave[13] = info, isave[14] = ifun, isave[15] = iword;
    isave[16] = (int32_t)std::min<int64_t>(nfree_g, INT32_MAX);
    isave[17] = (int32_t)std::min<int64_t>(nglob - nfree_g, INT32_MAX);
    isave[18] = (int32_t)std::min<int64_t>(ileave_g, INT32_MAX);
    isave[19] = (int32_t)std::min<int64_t>(nenter_g, INT32_MAX);
    dsave[0] = theta, dsave[1] = fold, dsave[2] = tol, dsave[3] = dnorm, dsave[4] = epsmch;
    dsave[5] = cpu1, dsave[6] = cachyt, dsave[7] = sbtime, dsave[8] = lnscht, dsave[9] = time1;
    dsave[10] = gd, dsave[11] = stpmx, dsave[12] = sbgnrm, dsave[13] = stp, dsave[14] = gdold;
    dsave[15] = dtd;
  }
  void finish(Mainlb &L) {  // :892-902
    MAINLB_VIEW(L);
    const double time = now_s() - time1;
    if (!quiet) {
      std::vector<double> xf;
      if (ipr >= 100 && !lbh::str60_pre(task, "ERROR")) xf = host_vec(x);
      rep.prn3lb(nglob, *f, task, ipr, info, iter, nfgv, nintol, nskip,
                 (int)std::min<int64_t>(nglob - nfree_g, INT32_MAX), sbgnrm, time, nseg, word,
                 iback, stp, xstep, err_k, cachyt, sbtime, lnscht, xf.empty() ? nullptr : xf.data());
    }
    save_locals(L);
  }
  void refresh(Mainlb &L) {
    MAINLB_VIEW(L);
    info = 0, col = 0, head = 1, theta = 1.0, iupdat = 0, updatd = false;
    nrefresh++;
    live_col = 0;
    pend.on = 0, pend.impl = 0;  // the memory is dropped, an uncommitted pair with it
  }

  // x = t, g = r (:568-569, :736-737).  Ping-pong buffers: the previous iterate still sits in its
  // own pair, which simply becomes the pair the caller is pointed at again.
  int restore_iterate(Mainlb &L) {
    restored_xg = true;
    if (pp && (t == xb[0] || t == xb[1])) {
      pp_cur = t == xb[0] ? 0 : 1;
      L.x = xb[pp_cur], L.g = gb[pp_cur];
      return 0;
    }
    if (L.x != t) HIPCHK(hipMemcpyAsync(L.x, t, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    if (L.g != r) HIPCHK(hipMemcpyAsync(L.g, r, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    return 0;
  }

  // task = 'START' (:430-507): errclb, active, the first f,g request
  int phase_start(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    spec.valid = false, pend.on = 0, pend.impl = 0, d_impl = z_in_x = false, scan.ready = false;
    ls.deferred = false, defer_live = false, wl.pending = false;
    nrefresh = 0;
    live_head = 1, live_col = 0, cw_stale = 0, cw_hold = 0, cw_settled = 0;  // (no pair is stored: any layout bits may stay)
    sfv.valid = false, sfv_hot = false, eager.valid = false, spec_live_len = 0;
    spcand.valid = false, last_tsum = 0.0, last_dtm0 = 0.0, iter_seen = 0, spec_factor = 2.0, last_walk_nseg = 0;
    epsmch = sizeof(T) == 4 ? (double)std::numeric_limits<float>::epsilon()
                            : std::numeric_limits<double>::epsilon();
    time1 = now_s();
    col = 0, head = 1, theta = 1.0, iupdat = 0, updatd = false, iback = 0, itail = 0;
    iword = 0, nact = 0, ileave = 0, nenter = 0, fold = 0, dnorm = 0, cpu1 = 0, gd = 0;
    stpmx = 0, sbgnrm = 0, stp = 0, gdold = 0, dtd = 0, iter = 0, nfgv = 0, nseg = 0;
    nintol = 0, nskip = 0, ifun = 0, cachyt = 0, sbtime = 0, lnscht = 0, info = 0;
    nfree_g = nglob, nenter_g = 0, ileave_g = 0;
    index_valid = false;
    tol = factr * epsmch;
    std::memcpy(word, "---", 4);
    prjctd = cnstnd = false, boxed = true;
    if (ipr >= 1 && !rep.itf) rep.itf = std::fopen(itfile_name.c_str(), "w");
    // errclb :1601-1643
    err_k = 0;
    if (nglob <= 0) lbh::str60_set(task, "ERROR: N <= 0");
    if (m <= 0) lbh::str60_set(task, "ERROR: M <= 0");
    if (factr < 0.0) lbh::str60_set(task, "ERROR: FACTR < 0");
    lbk::launch_errclb<T>(q, n, row0, l, u, nbd);
    CHK(fetch(0, 0, 5));
    {
      // uniform bounds (this rank's rows; every rank decides for itself: only loads are affected)
      ub_mask = 0;
      ub_tab = lbk::BoundTables{};
      // (errclb's answers before the probes below reuse h_res)
      const int64_t k6_ = (int64_t)h_res[0], k7_ = (int64_t)h_res[1];
      const bool uni_l = h_res[2] == 0.0, uni_u = h_res[3] == 0.0, uni_nb = h_res[4] == 0.0;
      if (ub_on) {
        T lu0[2];
        int32_t nb0 = 0;
        HIPCHK(hipMemcpy(&lu0[0], l, sizeof(T), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&lu0[1], u, sizeof(T), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&nb0, nbd, sizeof(int32_t), hipMemcpyDeviceToHost));
        // (h_res[2..4] are maxima over ALL ranks: a rank whose own rows are uniform but another's
        //  are not simply keeps streaming -- harmless)
        for (int j = 0; j < 8; ++j) ub_tab.l[j] = (double)lu0[0], ub_tab.u[j] = (double)lu0[1];
        ub_tab.nl = ub_tab.nu = 1, ub_tab.nb0 = nb0;
        if (uni_l) ub_mask |= 1;
        if (uni_u) ub_mask |= 2;
        if (uni_nb && nb0 >= 0 && nb0 <= 3) ub_mask |= 4;
        // few-valued l / u: build the tables by probing (every quantity reduced over the ranks: all ranks
        // hold the same tables after the same number of passes).  Only for valid input (errclb found nothing).
        if (dict_on && !(uni_l && uni_u) && k6_ == 0 && k7_ == 0) {
          lbk::BoundTables tb{};
          bool ok = true;
          auto member = [](const double *tab, int cnt, double v) {
            for (int j = 0; j < cnt; ++j)
              if (std::memcmp(&tab[j], &v, sizeof(double)) == 0) return true;
            return false;
          };
          for (int trip = 0; trip < 18 && ok; ++trip) {
            lbk::launch_dict_probe<T>(q, n, l, u, tb);
            CHK(fetch(2, 2, 0));
            const double cl = h_res[0], cu = h_res[1], vl = h_res[2], vu = h_res[3];
            if (cl == 0.0 && cu == 0.0) break;
            if (trip == 17) ok = false;
            if (cl > 0.0) {
              if (tb.nl == 8 || member(tb.l, tb.nl, vl)) ok = false;  // a 9th value, or one that == cannot find (NaN)
              else tb.l[tb.nl++] = vl;
            }
            if (cu > 0.0) {
              if (tb.nu == 8 || member(tb.u, tb.nu, vu)) ok = false;
              else tb.u[tb.nu++] = vu;
            }
          }
          if (ok && tb.nl >= 1 && tb.nu >= 1) {
            for (int j = tb.nl; j < 8; ++j) tb.l[j] = tb.l[0];
            for (int j = tb.nu; j < 8; ++j) tb.u[j] = tb.u[0];
            tb.nb0 = nb0;
            ub_tab = tb;
            ub_mask = 1 | 2 | lbk::UB_DICT;
            lbk::launch_nbd_pack_dict<T>(q, n, nbd, l, u, ub_tab, nbd8);
            nbd8_src = nbd;
          }
        }
        char host[192];
        std::memset(host, 0, sizeof host);
        for (int k = 0; k < 8; ++k) {  // (fp64: the whole 64 bytes; fp32: the first 32 -- every lane reads from the start)
          const T lv = (T)ub_tab.l[k], uv = (T)ub_tab.u[k];
          std::memcpy(host + k * sizeof(T), &lv, sizeof(T));
          std::memcpy(host + 64 + k * sizeof(T), &uv, sizeof(T));
        }
        if (sizeof(T) == 4 && !(ub_mask & lbk::UB_DICT))  // (uniform fp32: the value fills the buffer as before)
          for (int k = 8; k < 16; ++k) {
            const T lv = (T)ub_tab.l[0], uv = (T)ub_tab.u[0];
            std::memcpy(host + k * sizeof(T), &lv, sizeof(T));
            std::memcpy(host + 64 + k * sizeof(T), &uv, sizeof(T));
          }
        std::memset(host + 128, (int)(lbk::nb_t)nb0, 64);
        HIPCHK(hipMemcpy(ub_buf, host, sizeof host, hipMemcpyHostToDevice));
        ub_l = l, ub_u = u, ub_nbd = nbd;
      }
      h_res[0] = (double)k6_, h_res[1] = (double)k7_;
    }
    {
      const int64_t k6 = (int64_t)h_res[0], k7 = (int64_t)h_res[1];
      if (k6 > 0 || k7 > 0) {
        if (k6 > k7) {
          lbh::str60_set(task, "ERROR: INVALID NBD");
          info = -6, err_k = k6;
        } else {
          lbh::str60_set(task, "ERROR: NO FEASIBLE SOLUTION");
          info = -7, err_k = k7;
        }
      }
    }
    if (lbh::str60_pre(task, "ERROR")) {
      if (!quiet)
        rep.prn3lb(nglob, *f, task, ipr, info, iter, nfgv, nintol, nskip, nact, sbgnrm, 0.0,
                   nseg, word, iback, stp, xstep, err_k, cachyt, sbtime, lnscht);
      return done(flow);
    }
    if (!quiet) rep.prn1lb(nglob, m, ipr, epsmch);
    if (ipr > 100) {  // :2404-2408
      rep.vec_a4("L =", host_vec(l).data(), n);
      rep.vec_a4("X0 =", host_vec(x).data(), n);
      rep.vec_a4("U =", host_vec(u).data(), n);
    }
    lbk::launch_active<T>(q, n, x, l, u, nbd, iwhere, wasfree);  // :965-1040
    iw_dirty = 1.0;
    CHK(fetch(4, 0, 0));
    prjctd = h_res[0] > 0.0;
    cnstnd = h_res[1] > 0.0;
    boxed = h_res[2] == 0.0;
    if (!quiet) rep.active_msgs(ipr, prjctd, cnstnd, (long long)h_res[3]);
    if (prevfree) HIPCHK(hipMemsetAsync(prevfree, 1, (size_t)n, stream));
    lbh::str60_set(task, "FG_START");
    save_locals(L);
    return done(flow);
  }

  void phase_restore(Mainlb &L) {
    MAINLB_VIEW(L);
    // restore :511-550
    prjctd = lsave[0], cnstnd = lsave[1], boxed = lsave[2], updatd = lsave[3];
    nintol = isave[0], iback = isave[3], nskip = isave[4], head = isave[5], col = isave[6];
    itail = isave[7], iter = isave[8], iupdat = isave[9], nseg = isave[11], nfgv = isave[12];
    info = isave[13], ifun = isave[14], iword = isave[15];
    nfree = isave[16], nact = isave[17], ileave = isave[18], nenter = isave[19];
    (void)nfree, (void)nact, (void)ileave, (void)nenter;
    theta = dsave[0], fold = dsave[1], tol = dsave[2], dnorm = dsave[3], epsmch = dsave[4];
    cpu1 = dsave[5], cachyt = dsave[6], sbtime = dsave[7], lnscht = dsave[8], time1 = dsave[9];
    gd = dsave[10], stpmx = dsave[11], sbgnrm = dsave[12], stp = dsave[13], gdold = dsave[14];
    dtd = dsave[15];
  }

  // where the call re-enters mainlb (:552-577); for the first trial point of a line search this
  // is also its evaluation (update_scan_kernel run speculatively)
  int phase_entry(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    nrpre.valid = false;
    // value of a deferred built-in objective: one more sum in front of this call's first fetch
    if (f_pending) {
      f_pending = false;
      if (lbh::str60_pre(task, "FG")) {
        fo = 1;
      } else {  // (not an f,g return after all: just bring the value over)
        CHK(fetch(1, 0, 0));
        *f = f_scale * h_res[0];
      }
    }
    // a deferred line-search set-up lands with this call's first fetch -- if this IS the evaluation it
    // asked for; any other task drops it (the sums stay unread)
    const bool was_deferred = ls.deferred;
    const bool landing = ls.deferred && lbh::str60_pre(task, "FG_LN");
    if (!landing) ls.deferred = false, defer_live = false, wl.pending = false;
    if (lbh::str60_pre(task, "FG_LN")) {
      compute_pg = false, prelims = false;
      spec.valid = false;
      spcand.valid = false;
      // (landing: the set-up has not run; its step is the unit step, this is its first trial)
      const double stp_here = landing ? 1.0 : stp;
      // ... and the SECOND trial (an interpolated step after a rejected first one) is accepted nearly always: run
      // as lnsrlb_eval it costs 0.8 ms at n = 1e8 and the NEW_X entry that follows needs the full pass anyway
      // (2.6 ms + a host sync: profiles/round5_i_iter_timeline.txt, iterations 14 / 15); a third trial is a
      // backtracking search, evaluated the cheap way
      const bool first_trial = landing || ifun == 1 || (spec_trial2_on && ifun == 2);
      // First trial of a line search on a bounded problem: it is accepted far more often than
      // not, so evaluate it with the pass that matupd + the next cauchy scan would run anyway
      // (read-only with the pair pending); g'd and |proj g| are two of its sums.  Contexts
      // that mirror the reference's arrays at every return keep iwhere untouched until the
      // update is real (iwhere_update_kernel at the NEW_X entry).
      // Unconstrained problems (two_pass): the same pass -- every row is free, its p = W'd is
      // W'Z r itself (r = -g, c = 0), and the new pair needs no copy pass of its own.
      if ((cnstnd || two_pass) && first_trial && (!wide() || wide_fused())) {
        const int store_iw = (flags & LBFGSB_F_MIRROR_INDEX) ? 0 : 1;
        int c2, h2, it2;  // matupd's pointer update (:2303-2309), as if this trial is accepted
        if (iupdat + 1 <= m) {
          c2 = iupdat + 1, h2 = head, it2 = (head + iupdat - 1) % m + 1;
        } else {
          c2 = col, it2 = itail % m + 1, h2 = head % m + 1;
        }
        const int MCo = lbk::maxc_stride(c2 - 1);
        const int NX = lbk::update_scan_extra(c2 - 1, nr_flag(c2));
        clk_begin(1);
        q.res_off = fo;
        // (the MC = 20 instantiation with the new-row sums has no registers for the hand-over)
        const double chi = (nr_flag(c2) && lbk::maxc_for(c2 - 1) > 10) ? -1.0 : spec_hi(cnstnd);
        lbk::launch_update_scan<T>(q, n, x, lk(l), uk(u), nbk(), g, r, d_src(), d_impl ? 1 : 0, stp_here, iwhere,
                                   (T *)nullptr, Wc(), h2, c2, it2, 0, store_iw, nr_flag(c2), chi,
                                   sp_keys, sp_idx, SPEC_CAP, sp_count, ub_mask);
        q.res_off = 0;
        clk_end(1);
        spcand.valid = false;
        if (chi >= 0.0) CHK(spec_queue(x, l, u, g, h2, c2, stp_here));
        // freev (:1980-2059) and formk's patch sums (:1801-1851) of the NEXT iteration, speculatively: if this
        // trial point is accepted and the walk that follows fixes no row, iwhere is final as this pass has
        // just left it -- the counting pass (without its wasfree stores), the sort of its list and the
        // patch are queued right here and their sums come with this fetch (sfv.*; used in phase_cauchy_freev).
        // Only while the free set is changing from iteration to iteration (sfv_hot); otherwise the regular
        // route finds "nothing changed" without any pass at all.
        sfv.valid = false;
        const int sf_upcl = c2 - 1;
        const bool do_sfv = spec_freev_on && sfv_hot && cnstnd && store_iw && index_valid && !index && two_pass &&
                            c2 <= two_pass_maxcol && sf_upcl > 0 && lbk::maxc_for(sf_upcl) <= 20 &&
                            !(nr_flag(c2) && sf_upcl > q.tune.split_from) &&  // (never behind a SPLIT update pass: its
                            // merge kernel writes d_res directly, the chain's finalize would become the tail of the
                            // stream and the fetch would take stale update-pass sums from the host mirror)
                            chi < 0.0 &&
                            print_level < 99 &&
                            !(flags & LBFGSB_F_PARALLEL_GCP);
        if (do_sfv) {
          const int par = fv_parity;
          q.res_off = SPEC_OFF;
          q.hold_fin = fold_fin;  // (its four sums ride with the patch's finalize: one launch less)
          lbk::launch_freev_count(q, n, iwhere, wasfree, d_chg, CHG_CAP, d_fcount, par, 0);
          q.hold_fin = false;
          fv_parity ^= 1;
          lbk::launch_sort_u32_small_dev(q, d_chg, d_fcount + (par & 1));
          q.res_off = SPEC_OFF + 4;
          lbk::launch_formk_patch_dev<T>(q, d_chg, d_fcount + (par & 1), (uint32_t)lbk::small_sort_cap(), Wc(), h2,
                                         sf_upcl);
          q.res_off = 0;
          spec_live_len = 4 + 2 * sf_upcl * sf_upcl + sf_upcl + 1;
          sfv.parity = par & 1, sfv.upcl = sf_upcl, sfv.head = h2, sfv.col = c2, sfv.iter = iter + 1, sfv.x = x;
        }
        CHK(fetch(fo + 4 * MCo + 9 + NX, 1, 1));
        if (do_sfv) {
          const int nl = spec_live_len;
          spec_live_len = 0;
          const double *S = h_res + SPEC_OFF;
          for (int j = 0; j < 4; ++j) sfv.cnt[j] = S[j];
          sfv.loc3 = h_loc[SPEC_OFF + 3];
          sfv.P.assign(S + 4, S + nl - 1);
          sfv.served = S[nl - 1] == 0.0;  // (summed over the ranks: every rank's list fitted the chain)
          sfv.valid = true;
        }
        t_mid0 = now_s(), t_mark = t_mid0;
        if (chi >= 0.0) CHK(spec_land(c2, chi));
        if (fo) *f = f_scale * h_res[0];
        const double *R = h_res + fo;
        if (store_iw) iw_dirty += R[4 * MCo + 8];  // (the pass stored the entries that changed)
        L.spec_iw_changed = store_iw ? R[4 * MCo + 8] : 0.0;
        gd = R[4 * MCo + 7];
        spec_sbgnrm = R[4 * MCo + 10 + NX];
        std::memcpy(spec.res.data(), R, sizeof(double) * (4 * MCo + 11 + NX));
        spec.valid = true;  // dropped below unless dcsrch accepts this point
        spec.x = x, spec.g = g, spec.stp = stp_here, spec.head = h2, spec.col = c2, spec.itail = it2;
        tbrk_valid = false;
      } else {
        // g.d for the line search and, speculatively, |proj g| for the NEW_X return
        CHK(ensure_d(x));
        q.res_off = fo;
        lbk::launch_lnsrlb_eval<T>(q, n, x, l, u, nbd, g, d);
        q.res_off = 0;
        CHK(fetch(fo + 1, 0, 1));
        if (fo) *f = f_scale * h_res[0];
        gd = h_res[fo];
        spec_sbgnrm = h_res[fo + 1];
      }
      if (landing) CHK(land_deferred(L));
    } else if (lbh::str60_pre(task, "NEW_X")) {
      seg(1);
      compute_pg = false, prelims = false, linesearch = false;
      // The caller's l, u, nbd against the snapshot the passes over W read -- after the first iteration, then every
      // bcheck_every-th: the reference re-reads the arrays on every call (:1270-1330, :2594-2622, :2789-2816), so
      // an edit in place would take effect there; here it ends the run with an error instead of being ignored.
      // (At a NEW_X entry nothing deferred is in flight: the fetch is this call's own.)
      if (bcheck_every > 0 && (iter == 1 || iter % bcheck_every == 0)) {
        lbk::launch_bounds_verify<T>(q, n, l, u, nbd, nbd8, ub_mask, ub_tab);
        CHK(fetch(1, 0, 0));
        nbounds_checks++;
        if (h_res[0] != 0.0) {
          lbh::str60_set(task, "ERROR: BOUNDS CHANGED DURING RUN");
          info = -10;
          finish(L);
          return done(flow);
        }
      }
    } else if (!lbh::str60_pre(task, "FG_ST")) {
      if (lbh::str60_pre(task, "STOP")) {
        if (std::strncmp(task + 6, "CPU", 3) == 0) {  // :566-571
          CHK(restore_iterate(L));
          HIPCHK(hipStreamSynchronize(stream));
          // (a line search whose set-up was still deferred has not stored fold = f yet, :2237)
          if (was_deferred) fold = defer_f0;
          *f = fold;
        }
        finish(L);
      } else {
        lbh::str60_set(task, "FG_START");
        save_locals(L);
      }
      return done(flow);
    }
    return 0;
  }

  // LBFGSB_F_DEFER_LNSRCH: the sums of the previous call's storing pass have arrived (h_res[DEFER_OFF..+4)),
  // together with the evaluation of the trial point that call had already put in place.  Go back to the
  // iterate and let the line-search set-up run as it would have in that call; if it asks for exactly
  // this trial point -- the usual case -- phase_linesearch continues with the evaluation at once
  // (Mainlb::spec_holds), otherwise (backtracking step, ascent direction) the corrected request goes
  // back to the caller: one more 'FG_LNSRCH', the wasted evaluation is not counted in nfgv.
  int land_deferred(Mainlb &L) {
    MAINLB_VIEW(L);
    const double D[4] = {h_res[DEFER_OFF], h_res[DEFER_OFF + 1], h_res[DEFER_OFF + 2], h_res[DEFER_OFF + 3]};
    ls.deferred = false, defer_live = false;
    L.f_trial = *f, L.gd_trial = gd, L.sbg_trial = spec_sbgnrm;
    L.landing = true, L.spec_holds = false;
    if (pp) pp_cur ^= 1, L.x = xb[pp_cur], L.g = gb[pp_cur];  // the iterate's pair again
    *f = defer_f0;
    linesearch = true, prelims = false, compute_pg = false;
    int info_sub = 0;
    const bool uphill = D[0] > 0.0 && D[1] > 0.0;  // subsm's backtracking branch (:2828)
    if (uphill) {
      // the branch reads iwhere as the walk left it at the ITERATE; the pass that evaluated the trial
      // point has meanwhile stored the entries that change there.  The post-scan status is a function of
      // the row's own x, g and bounds (:1284-1291), the walk's fixes are on file: put both back
      spec.valid = false;
      iw_dirty -= L.spec_iw_changed;
      CHK(restore_iterate(L));  // (classic entry: x = t, g = r)
      if (cnstnd && L.spec_iw_changed > 0.0) {
        lbk::launch_iwhere_update<T>(q, n, L.x, L.l, L.u, L.nbd, L.g, iwhere);
        tbrk_valid = false;
        CHK(apply_walk_fixes());
      }
    }
    if (wl.pending)
      CHK(wide_land(L.x, L.l, L.u, L.nbd, L.g, D, iword));
    else
      CHK(subspace_land(L.x, L.l, L.u, L.nbd, L.g, D, iword, info_sub));
    return 0;  // (phase_linesearch counts the evaluation as wasted if its set-up asks for another point)
  }

  // first projected gradient (:579-596)
  int phase_first_projgr(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    nfgv = 1;
    q.res_off = fo;
    lbk::launch_projgr<T>(q, n, x, l, u, nbd, g);
    q.res_off = 0;
    CHK(fetch(fo, 0, 1));
    if (fo) *f = f_scale * h_res[0];
    sbgnrm = h_res[fo];
    if (!quiet) rep.iterate0(ipr, iter, nfgv, *f, sbgnrm);
    if (sbgnrm <= pgtol) {
      lbh::str60_set(task, "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL");
      finish(L);
      return done(flow);
    }
    return 0;
  }

  // generalized Cauchy point + freev (:599-646)
  // freev's counting pass (:1980-2059), in two halves so that another pass can be queued between
  // the launch and the one host sync that brings its three counts (h_res[0..2])
  int freev_launch(bool track, bool park_finalize = false) {
    if (prevfree)
      HIPCHK(hipMemcpyAsync(prevfree, wasfree, (size_t)n, hipMemcpyDeviceToDevice, stream));
    q.hold_fin = park_finalize && fold_fin;  // (the patch chain that follows finalizes both)
    lbk::launch_freev_count(q, n, iwhere, wasfree, track ? d_chg : nullptr, CHG_CAP, d_fcount, fv_parity);
    q.hold_fin = false;
    fv_parity ^= 1;
    index_valid = true;
    iw_dirty = 0.0;
    return 0;
  }
  bool freev_land(bool track, bool updatd) {  // -> wrk (:2057)
    // (the length of THIS rank's changed-row list: its own sum, before the reduction over ranks)
    chg_local = track ? (uint32_t)std::min<double>(h_loc[3], 4294967295.0) : 0;
    nfree_g = (int64_t)h_res[0];
    if (track) {
      nenter_g = (int64_t)h_res[1];
      ileave_g = nglob + 1 - (int64_t)h_res[2];
    } else {
      nenter_g = 0;
      ileave_g = nglob + 1;
    }
    return (ileave_g < nglob + 1) || (nenter_g > 0) || updatd;
  }

  int phase_cauchy_freev(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    if (ipr >= 99) std::fprintf(rep.out, "\n\nITERATION %5d\n", iter + 1);
    iword = -1;
    ls.ready = false;
    ls.x_is_z = false;
    ls_do_stpmx = cnstnd && iter != 0;
    ls_unit_step = !(iter == 0 && !boxed);  // lnsrlb :2228-2232
    xmut = pp ? xb[1 - pp_cur] : x;  // where this iteration's trial points go
    // this trip's operands (an unconstrained problem never calls cauchy(), and with ping-pong buffers
    // the iterate changes place from one iteration to the next)
    cx = x, cl = l, cu = u, cg = g, cnbd = nbd;
    if (!cnstnd && col > 0) {  // :607-611  (z = x, kept in functional form)
      gcp = Gcp{};
      gcp.copy_x = true;
      z_valid = false, z_in_x = false;
      wrk = updatd;
      nseg = 0;
      pre_valid = false;
      // no walk: the update pass's p = W'd over all rows IS W'Z r (subspace_closed_form with
      // tsum = 0 and no cmprlb term), its new-row sums need no correction
      closed_ok = false;
      nrc_clear();
      if (scan.ready) {
        for (int j = 0; j < col; ++j) {
          p_fin[j] = scan.p[j];
          p_fin[col + j] = theta != 1.0 ? theta * scan.p[col + j] : scan.p[col + j];  // :1337
        }
        closed_ok = true;
        scan.ready = false;
      }
    } else {
      cpu1 = now_s();
      CHK(cauchy(x, l, u, nbd, g, theta, col, head, sbgnrm, epsmch, nseg, info));
      if (info != 0) {  // :620-635
        if (ipr >= 1)
          std::fprintf(rep.out,
                       "\n Singular triangular system detected;\n   refresh the lbfgs "
                       "memory and restart the iteration.\n");
        refresh(L);
        cachyt += now_s() - cpu1;
        return again(flow);
      }
      // freev :1980-2059 (counts; the lists only when mirroring Index)
      // (two-pass iteration: no cmprlb pass if the closed form applies -- decided for good
      //  once nfree is known, below)
      const bool closed_cand = two_pass && closed_ok && col > 0 && col <= two_pass_maxcol &&
                               (!updatd || (nrpre.valid && nrpre.col == col));
      const bool track = iter > 0 && cnstnd;  // freev looks for entering/leaving rows (:2012)
      pre_valid = false;
      // (m > 32: the same -- nothing downstream of freev rides on its fetch there, closed form or not)
      if (track && (closed_cand || wide()) && index_valid && !index && iw_dirty == 0.0) {
        // no iwhere entry has changed since the last freev: nobody enters, nobody leaves, nfree stands
        // -- neither the counting pass nor a host sync (iw_dirty is a sum over all ranks)
        nfreev_skipped++;
        chg_local = 0;
        cachyt += now_s() - cpu1;
        nintol += nseg;
        nenter_g = 0;
        ileave_g = nglob + 1;
        wrk = updatd;
        if (ipr >= 99) {  // :2023-2057
          std::fprintf(rep.out, " %11lld  variables leave; %11lld  variables enter\n", 0ll, 0ll);
          std::fprintf(rep.out, " %11lld  variables are free at GCP %11d\n", (long long)nfree_g, iter + 1);
        }
        seg(3);
        return 0;
      }
      if (sfv.valid && sfv.served && track && closed_cand && !index && fixlist.empty() && !fix_overflow &&
          sfv.iter == iter && sfv.x == x && sfv.col == col && sfv.head == head && sfv.upcl == (updatd ? col - 1 : col)) {
        // the speculative chain behind the accepted trial point holds: the walk fixed nothing, so iwhere is what
        // that counting pass saw.  Its counts are freev's, its patch is formk's; wasfree is brought up to date
        // from its list (<= the chain's capacity on every rank: sfv.served) -- no pass, no host sync here
        sfv.valid = false;
        nsfv_used++;
        if (sfv.cnt[3] > 0.0)
          lbk::launch_freev_apply(q, d_chg, d_fcount + sfv.parity, (uint32_t)lbk::small_sort_cap(), wasfree);
        iw_dirty = 0.0;
        h_res[0] = sfv.cnt[0], h_res[1] = sfv.cnt[1], h_res[2] = sfv.cnt[2];
        h_loc[3] = sfv.loc3;
        cachyt += now_s() - cpu1;
        nintol += nseg;
        wrk = freev_land(track, updatd);
        sfv_hot = sfv.cnt[3] > 0.0;
        eager.valid = true, eager.upcl = sfv.upcl, eager.head = sfv.head, eager.P = sfv.P;
        seg(3);
        return 0;
      }
      sfv.valid = false;
      const bool will_eager = eager_on && track && closed_cand && (updatd ? col - 1 : col) > 0 && !wide();
      CHK(freev_launch(track, will_eager));
      // formk's patch sums (:1801-1851) do not depend on anything the host still has to decide either -- only
      // on the list freev's pass has just written and on W: in the two-pass iteration (no cmprlb launch
      // below) the sort of the list and the patch kernel are queued right behind the counting pass with
      // the list's length read on the device, and ONE fetch brings freev's counts and the patch
      // (eager.*; lists longer than the one-workgroup sort serves fall back to the ordinary route in
      // formk_incremental).  Same kernels, same order of the sums: bit-identical to that route.
      eager.valid = false;
      int neager = 0;
      {
        const int upcl = updatd ? col - 1 : col;
        if (will_eager) {
          const uint32_t *cnt_ptr = d_fcount + ((fv_parity ^ 1) & 1);  // the counter freev_launch just used
          lbk::launch_sort_u32_small_dev(q, d_chg, cnt_ptr);
          q.res_off = 4;
          lbk::launch_formk_patch_dev<T>(q, d_chg, cnt_ptr, (uint32_t)lbk::small_sort_cap(), Wc(), head, upcl);
          q.res_off = 0;
          neager = 2 * upcl * upcl + upcl + 1;
          eager.upcl = upcl, eager.head = head;
        }
      }
      // the cmprlb pass does not depend on freev's counts: launch it now and fetch both
      // sets of sums with ONE host sync (it is wasted only if no variable is free)
      int npre = 0;
      double p_walk[2 * lbk::MAXM];  // (wa(1:2m) as the walk left it: cmprlb_coef puts M c there)
      if (col > 0 && !closed_cand && !wide()) {
        lbk::Coef cf;
        bool plain;
        std::memcpy(p_walk, wa8m.data(), sizeof(double) * 2 * col);
        if (cmprlb_coef(col, theta, cnstnd, cf, plain)) {
          const bool newrow = updatd && col <= lbk::MAXM;  // updatd implies wrk
          CHK(ensure_d(x));
          q.res_off = 4;
          clk_begin(0);
          lbk::launch_cmprlb_wtv<T>(q, n, x, g, gcp.tsum, iwhere, W(), head, col, theta, cf,
                                    newrow ? 1 : 0, r, d, pend);
          clk_end(0);
          q.res_off = 0;
          npre = (newrow ? 6 : 2) * lbk::maxc_for(col);
        }
      }
      CHK(fetch(4 + npre + neager, 0, 0));
      if (neager) {  // (never together with npre: closed_cand)
        eager.P.assign(h_res + 4, h_res + 4 + neager - 1);
        eager.valid = h_res[4 + neager - 1] == 0.0;  // (summed over the ranks: every rank's list was served)
        neager_served += eager.valid;
      }
      if (npre) {
        std::memcpy(pre_res, h_res + 4, sizeof(double) * npre);
        pre_valid = true;
      }
      cachyt += now_s() - cpu1;
      nintol += nseg;
      wrk = freev_land(track, updatd);
      // no free variable: the reference goes straight to the line search (:648-651), wa(1:2m) stays cauchy's p
      if (npre && nfree_g == 0) std::memcpy(wa8m.data(), p_walk, sizeof(double) * 2 * col);
      sfv_hot = track && (nenter_g > 0 || ileave_g < nglob + 1);  // the free set is moving: speculate next time
      if (ipr >= 99) {  // :2023-2057
        if (iter > 0 && cnstnd) {
          if (ipr >= 100 && chg_local > 0 && chg_local <= CHG_CAP) {
            // this rank's rows that changed status: leaving rows in ascending order (the scan
            // of Index(1:nfree)), entering rows in descending order (the active part of Index
            // is filled from the back)
            std::vector<uint32_t> ch(chg_local);
            (void)hipMemcpyAsync(ch.data(), d_chg, chg_local * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, stream);
            (void)hipStreamSynchronize(stream);
            std::vector<int64_t> lv, en;
            for (uint32_t e : ch) ((e & 0x80000000u) ? lv : en).push_back((int64_t)(e & 0x7fffffffu));
            std::sort(lv.begin(), lv.end());
            std::sort(en.begin(), en.end(), std::greater<int64_t>());
            for (int64_t k : lv)
              std::fprintf(rep.out, " Variable %11lld  leaves the set of free variables\n",
                           (long long)(row0 + k + 1));
            for (int64_t k : en)
              std::fprintf(rep.out, " Variable %11lld  enters the set of free variables\n",
                           (long long)(row0 + k + 1));
          }
          std::fprintf(rep.out, " %11lld  variables leave; %11lld  variables enter\n",
                       (long long)(nglob + 1 - ileave_g), (long long)nenter_g);
        }
        std::fprintf(rep.out, " %11lld  variables are free at GCP %11d\n", (long long)nfree_g,
                     iter + 1);
      }
      if (index)
        lbk::launch_freev_lists(q, n, iwhere, prevfree, (iter > 0 && cnstnd) ? 1 : 0, index,
                                indx2, scan_tmp);
    }
    seg(3);
    return 0;
  }

  // formk + cmprlb + subsm (:648-712)
  int phase_subspace(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    if (nfree_g == 0 || col == 0) {
      // skip the subspace minimization :648-651: the line search starts from z = xcp
      CHK(commit_pending(g, col, head));
      CHK(ensure_z(x, l, u, g));
    } else {
      cpu1 = now_s();
      // incremental WN1 (new row / column sums ride in the cmprlb pass, status changes are patched):
      // every col the fused kernels take (round 3: col <= 20; beyond that formk ran from scratch in every
      // iteration -- 30 ms of the 45 at m = 32, n = 5e7)
      const bool incr = wrk && col <= lbk::MAXM && !wide();
      if (wrk && !incr) {
        // (m > 32: the new pair's row alone when no row changed status, solver_wide.inl)
        bool incr_done = false;
        // (from the FIRST formk of a run on, as the reference: a formk it skipped -- no free variable -- leaves
        //  that pair's row unwritten there too, its K fails and the memory is refreshed: fuzz 80740)
        if (wide() && wide_incr_on) {
          // (the new pair's row and column: from the update pass if it carried them -- corrected for the rows
          //  the walk fixed, as in subspace() -- else from two masked columns)
          std::vector<double> nrp;
          if (updatd && nrpre.valid && nrpre.col == col) {
            nrp.assign((size_t)4 * col, 0.0);
            for (int j = 0; j < col; ++j) {
              nrp[0 * col + j] = nrpre.t[0][j] - nrc[0][j];
              nrp[1 * col + j] = nrpre.t[1][j] + nrc[1][j];
              nrp[2 * col + j] = nrpre.t[2][j] + nrc[2][j];
              nrp[3 * col + j] = nrpre.t[3][j] - nrc[3][j];
            }
          }
          CHK(wide_formk_incr(col, head, updatd, iupdat, incr_done, nrp.empty() ? nullptr : nrp.data()));
        }
        if (incr_done)
          formk_factor(col, theta, info);
        else
          CHK(formk(col, head, theta, info));
      }
      if (info != 0) {  // :666-682
        if (ipr >= 1)
          std::fprintf(rep.out,
                       "\n Nonpositive definiteness in Cholesky factorization in formk;\n   "
                       "refresh the lbfgs memory and restart the iteration.\n");
        refresh(L);
        sbtime += now_s() - cpu1;
        return again(flow);
      }
      // closed form (subspace() itself checks, once WN1 is up to date, that S'ZZ'S = S'S - S'AA'S
      // does not cancel)
      const bool closed = two_pass && closed_ok && col <= two_pass_maxcol && !pre_valid &&
                          (!updatd || (nrpre.valid && nrpre.col == col));
      if (wide()) {
        // (wide_subspace commits the pending pair: in its one-launch r pass, or in front of the unfused steps)
        const bool wclosed = wide_fused() && wide_closed_on && two_pass && closed_ok && !pre_valid &&
                             (!updatd || (nrpre.valid && nrpre.col == col)) && closed_form_safe(col);
        CHK(wide_subspace(x, l, u, nbd, g, theta, col, head, cnstnd, iword, info, wclosed));
      } else {
        // (iwhere is final for this iteration: the moment to re-sort the tiles of W, if the policy wants it)
        cw_maybe_pack(head, col, nenter_g + (nglob + 1 - ileave_g));
        CHK(subspace(x, l, u, nbd, g, theta, col, head, cnstnd, iword, info, incr, updatd, iupdat,
                     pre_valid ? pre_res : nullptr, closed));
      }
      pre_valid = false;
      if (info == -1 || info == -2) {  // formk failed inside the fused pass (:666-682)
        if (ipr >= 1)
          std::fprintf(rep.out,
                       "\n Nonpositive definiteness in Cholesky factorization in formk;\n   "
                       "refresh the lbfgs memory and restart the iteration.\n");
        refresh(L);
        sbtime += now_s() - cpu1;
        return again(flow);
      }
      if (info != 0) {  // :694-710
        if (ipr >= 1)
          std::fprintf(rep.out,
                       "\n Singular triangular system detected;\n   refresh the lbfgs "
                       "memory and restart the iteration.\n");
        refresh(L);
        sbtime += now_s() - cpu1;
        return again(flow);
      }
      sbtime += now_s() - cpu1;
    }
    cpu1 = now_s();
    return 0;
  }

  // lnsrlb (:714-792): set-up, dcsrch, the next trial point or the accepted one
  int phase_linesearch(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    const double big = 1.0e10, ftol = 1.0e-3, gtol = 0.9, xtol = 0.1;
    bool ls_abort = false;
    // first call of this iteration's line search (or its deferred set-up, landing one call later)
    const bool setup_call = L.landing || !lbh::str60_pre(task, "FG_LN");
    if (setup_call && ls.deferred) {
      // LBFGSB_F_DEFER_LNSRCH: the storing pass has put the first trial point x = z in place; its sums
      // (dtd, g'd, stpmx, iword) are still on the device and come over with the next call's first
      // fetch -- no host sync in this call.  dsave / isave of THIS return do not describe the line
      // search yet (include/lbfgsb_hip.h)
      defer_f0 = *f;
      lbh::str60_set(task, "FG_LNSRCH");
      if (pp) pp_cur ^= 1, L.x = xb[pp_cur], L.g = gb[pp_cur];
      save_locals(L);
      return done(flow);
    }
    if (setup_call) {
      n_ls_setup++;
      const int do_stpmx = (cnstnd && iter != 0) ? 1 : 0;
      double stpmx_cand;
      if (ls.ready) {  // d, t, r, dtd, g'd came out of the subsm pass
        dtd = ls.dtd, gd = ls.gd, stpmx_cand = ls.stpmx;
      } else {
        lbk::launch_lnsrlb_begin<T>(q, n, z, x, g, l, u, nbd, d, pp ? (T *)nullptr : t, pp ? (T *)nullptr : r,
                                    do_stpmx);
        if (pp) t = x, r = g;  // (roles instead of copies)
        CHK(fetch(2, 1, 0));
        dtd = h_res[0], gd = h_res[1], stpmx_cand = h_res[2];
      }
      ls.ready = false;
      dnorm = std::sqrt(dtd);
      stpmx = big;
      if (cnstnd) stpmx = iter == 0 ? 1.0 : std::min(big, stpmx_cand);
      stp = (iter == 0 && !boxed) ? std::min(1.0 / dnorm, stpmx) : 1.0;
      fold = *f;
      ifun = 0;
      iback = 0;
      lbh::str60_set(csave, "START");
    }
    if (ifun == 0) {
      gdold = gd;
      if (gd >= 0.0) {  // :2247-2253
        if (!quiet)  // the reference prints this regardless of iprint (:2250)
          std::fprintf(rep.out, "  ascent direction in projection gd = %s\n",
                       lbr::flist(gd).c_str());
        info = -4;
        ls_abort = true;
      }
    }
    if (!ls_abort) {
      lbh::dcsrch(*f, gd, stp, ftol, gtol, xtol, 0.0, stpmx, csave, isave + 21, dsave + 16);
      xstep = stp * dnorm;
      if (!lbh::str60_pre(csave, "CONV") && !lbh::str60_pre(csave, "WARN")) {
        lbh::str60_set(task, "FG_LNSRCH");
        ifun++;
        nfgv++;
        iback = ifun - 1;
        const bool in_place = ls.x_is_z && ifun == 1 && stp == 1.0;  // x = z is already in place
        if (!in_place) {
          // (the set-up call writes this iteration's first trial point: xmut -- the caller's x, or the
          //  other buffer of a ping-pong pair; later calls are entered with x = the trial buffer)
          T *xt = setup_call ? xmut : x;
          bool stepped = false;
          CHK(ensure_d(xt, xt, stp, &stepped));  // (it still holds the rejected first trial point z)
          if (!stepped) lbk::launch_lnsrlb_step<T>(q, n, xt, z, d, t, stp);
        }
        ls.x_is_z = false;
        // (a landing set-up that asks for the point in place: that point HAS been evaluated, by the pass
        //  whose sums this call entered with -- they stay valid)
        L.spec_holds = L.landing && in_place;
        if (!L.spec_holds) {
          spec.valid = false;  // the trial point was not accepted
          spcand.valid = false;
        }
      } else {
        lbh::str60_set(task, "NEW_X");
      }
    } else {
      spec.valid = false;
      spcand.valid = false;
    }

    if (info != 0 || iback >= 20) {  // :734-769
      if (L.landing) L.landing = false, L.spec_holds = false, nredo++;
      CHK(ensure_d(setup_call ? xmut : x));  // (d, z as vectors before the trial point goes)
      CHK(restore_iterate(L));
      *f = fold;
      if (col == 0) {
        if (info == 0) {
          info = -9;
          nfgv--, ifun--, iback--;
        }
        lbh::str60_set(task, "ABNORMAL_TERMINATION_IN_LNSRCH");
        iter++;
        HIPCHK(hipStreamSynchronize(stream));
        finish(L);
        return done(flow);
      }
      if (ipr >= 1)
        std::fprintf(rep.out,
                     "\n Bad direction in the line search;\n   refresh the lbfgs memory and "
                     "restart the iteration.\n");
      if (info == 0) nfgv--;
      refresh(L);
      lbh::str60_set(task, "RESTART_FROM_LNSRCH");
      lnscht += now_s() - cpu1;
      prelims = linesearch = true;
      return again(flow);
    } else if (lbh::str60_pre(task, "FG_LN")) {
      // ping-pong buffers: the caller evaluates f, g at the OTHER pair from here on
      if (pp && setup_call) pp_cur ^= 1, L.x = xb[pp_cur], L.g = gb[pp_cur];
      if (L.landing && L.spec_holds) {
        // the request is the point this call was entered with: go on as its 'FG_LNSRCH' re-entry
        L.landing = false, L.spec_holds = false;
        *f = L.f_trial, gd = L.gd_trial, spec_sbgnrm = L.sbg_trial;
        return repeat(flow);
      }
      if (L.landing) L.landing = false, nredo++;  // (a corrected request: one more evaluation)
      save_locals(L);
      if (!(flags & (LBFGSB_F_NO_RETURN_SYNC | LBFGSB_F_DEFER_LNSRCH))) {
        const double t0 = now_s();
        HIPCHK(hipStreamSynchronize(stream));  // x is ready for the caller's f,g evaluation
        t_wait += now_s() - t0;
        nsync++;
      }
      return done(flow);
    } else {
      lnscht += now_s() - cpu1;
      iter++;
      sbgnrm = spec_sbgnrm;  // projgr (:781) was evaluated with g.d: x, g unchanged since
      switch (iword) {       // prn2lb :2438-2443
        case 0: std::memcpy(word, "con", 4); break;
        case 1: std::memcpy(word, "bnd", 4); break;
        case 5: std::memcpy(word, "TNT", 4); break;
        default: std::memcpy(word, "---", 4);
      }
      if (!quiet)
        rep.prn2lb(ipr, iter, nfgv, (int)std::min<int64_t>(nglob - nfree_g, INT32_MAX), sbgnrm,
                   nseg, word, iback, stp, xstep, *f);
      if (ipr > 100) {  // :2449-2452
        rep.vec_a4("X =", host_vec(x).data(), n);
        rep.vec_a4("G =", host_vec(g).data(), n);
      }
      save_locals(L);
      seg(0);
      return done(flow);
    }
    return 0;
  }

  // NEW_X re-entry: termination tests (:794-810)
  int phase_termination(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    // ---- NEW_X re-entry: termination tests :794-810 ----
    if (sbgnrm <= pgtol) {
      lbh::str60_set(task, "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL");
      finish(L);
      return done(flow);
    }
    double ddum = std::max(std::max(std::fabs(fold), std::fabs(*f)), 1.0);
    if ((fold - *f) <= tol * ddum) {
      lbh::str60_set(task, "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH");
      if (iback >= 10) info = -5;
      finish(L);
      return done(flow);
    }
    return 0;
  }

  // the m x m half of matupd (:2324-2346): shift of Sy, Ss once the memory is full, the new row of Sy and
  // column of Ss from the n-length sums, the two diagonal entries
  void matupd_small(int col, int iupdat, const double *sy_row, const double *ss_col, double ss_diag,
                    double dr) {
    lbh::Mat SY{sy.data(), m}, SS{ss.data(), m};
    if (iupdat > m) {  // :2324-2330
      for (int j = 0; j < col - 1; ++j) {
        for (int i = 0; i <= j; ++i) SS(i, j) = SS(i + 1, j + 1);
        for (int i = j; i < col - 1; ++i) SY(i, j) = SY(i + 1, j + 1);
      }
    }
    for (int j = 0; j < col - 1; ++j) {
      SY(col - 1, j) = sy_row[j];
      SS(j, col - 1) = ss_col[j];
    }
    SS(col - 1, col - 1) = ss_diag;
    SY(col - 1, col - 1) = dr;
  }

  // The BFGS update is skipped (:822-830) but the pass that evaluated the accepted point has run all the
  // same: the n-loop of the cauchy() that follows (:1270-1330) is a function of x, g, the bounds and the
  // columns of W, and the pass read every column that stays -- all of them while the memory is filling
  // (c2 = col + 1, same head); with the memory full it left out the OLDEST pair (it would have been
  // dropped), whose two products come from a one-column scan: 70 bytes per row instead of the 8 + 16 col + ...
  // of cauchy_scan_kernel over all columns.  f1, the breakpoint counts and bkmin do not depend on W at all.
  bool skip_reuse_on = true;  // option "skip_reuse"
  int64_t nskip_reused = 0;
  int scan_from_skipped(Mainlb &L) {
    MAINLB_VIEW(L);
    const int c2 = spec.col, nold = c2 - 1;
    const bool full = c2 == col;  // (the pass ran with head + 1)
    if (!(full ? (col == m && spec.head == head % m + 1) : (c2 == col + 1 && spec.head == head))) return 0;
    const int MCo = lbk::maxc_stride(nold);
    const int NX = lbk::update_scan_extra(nold, nr_flag(c2));
    const double *R = spec.res.data();
    const int shift = full ? 1 : 0;
    if (full) {
      const int MC1 = lbk::maxc_for(1);
      lbk::launch_cauchy_scan<T>(q, n, x, l, u, nbd, g, iwhere, tbrk, W(), head, 1);
      CHK(fetch(2 * MC1 + 4, 1, 0));
      scan.p[0] = h_res[0], scan.p[col] = h_res[MC1];
      tbrk_valid = true;
      if (flags & LBFGSB_F_MIRROR_INDEX) iw_dirty += 1.0;  // (the evaluation held its iwhere stores back)
    } else if ((flags & LBFGSB_F_MIRROR_INDEX) && R[4 * MCo + 8] > 0.0) {
      lbk::launch_iwhere_update<T>(q, n, x, l, u, nbd, g, iwhere), iw_dirty += 1.0;
    }
    for (int j = shift; j < col; ++j) {
      scan.p[j] = R[2 * MCo + 1 + j - shift];
      scan.p[col + j] = R[3 * MCo + 2 + j - shift];
    }
    scan.f1 = R[4 * MCo + 3], scan.nbreak = R[4 * MCo + 4];
    scan.nunb = R[4 * MCo + 5], scan.nunbnz = R[4 * MCo + 6];
    scan.bkmin = R[4 * MCo + 9 + NX];
    scan.ready = true;
    nskip_reused++;
    return 0;
  }

  // y, s and matupd (:812-857)
  int phase_update(Mainlb &L, Flow &flow) {
    MAINLB_VIEW(L);
    // ---- :812-834 ----
    double dr, ddum;
    if (stp == 1.0) {
      dr = gd - gdold;
      ddum = -gdold;
    } else {
      dr = (gd - gdold) * stp;
      ddum = -gdold * stp;
    }
    if (dr <= epsmch * ddum) {
      nskip++;
      updatd = false;
      if (debug_walk) std::fprintf(stderr, "[update] iter %d skipped: dr %g ddum %g\n", iter, dr, ddum);
      if (skip_reuse_on && spec.valid && spec.x == x && spec.g == g && spec.stp == stp && (!wide() || wide_fused()))
        CHK(scan_from_skipped(L));
      spec.valid = false;
      spcand.valid = false;
      if (ipr >= 1)
        std::fprintf(rep.out, "  ys=%s  -gs=%s BFGS update SKIPPED\n", lbr::fE(dr, 10, 3).c_str(),
                     lbr::fE(ddum, 10, 3).c_str());
      prelims = linesearch = true;
      return again(flow);
    }

    // ---- matupd :2291-2346 (pointer bookkeeping on the host) ----
    updatd = true;
    iupdat++;
    if (iupdat <= m) {
      col = iupdat;
      itail = (head + iupdat - 2) % m + 1;
    } else {
      itail = itail % m + 1;
      head = head % m + 1;
    }
    const int MCo = lbk::maxc_stride(col - 1);
    double rr;
    std::vector<double> wsy, wss;  // (m > 32: Sy's new row, Ss's new column from the tile passes)
    const bool unfused = wide() && !wide_fused();
    if (unfused) {
      CHK(ensure_d(x));
      CHK(wide_matupd(g, stp, head, col, wsy, wss, rr));
      spec.valid = false, pend.on = 0, scan.ready = false;
    } else if (cnstnd || two_pass) {
      // the next loop trip starts with cauchy: do its n-loop in the same pass over W --
      // unless that pass already ran as the evaluation of the accepted trial point
      const bool reuse = spec.valid && spec.x == x && spec.g == g && spec.stp == stp &&
                         spec.head == head && spec.col == col && spec.itail == itail;
      const int NX = lbk::update_scan_extra(col - 1, nr_flag(col));
      if (debug_walk) std::fprintf(stderr, "[update] iter %d stp %g reuse %d\n", iter, stp, (int)reuse);
      if (!reuse) sfv.valid = false;
      if (reuse) {
        std::memcpy(h_res, spec.res.data(), sizeof(double) * (4 * MCo + 11 + NX));
        if ((flags & LBFGSB_F_MIRROR_INDEX) && h_res[4 * MCo + 8] > 0.0)
          lbk::launch_iwhere_update<T>(q, n, x, l, u, nbd, g, iwhere), iw_dirty += 1.0;  // the pass held it back
      } else {
        clk_begin(1);
        const double chi = (nr_flag(col) && lbk::maxc_for(col - 1) > 10) ? -1.0 : spec_hi(cnstnd);
        lbk::launch_update_scan<T>(q, n, x, lk(l), uk(u), nbk(), g, r, d_src(), d_impl ? 1 : 0, stp, iwhere,
                                   (T *)nullptr, Wc(), head, col, itail, 0, 1, nr_flag(col), chi,
                                   sp_keys, sp_idx, SPEC_CAP, sp_count, ub_mask);
        clk_end(1);
        spcand.valid = false;
        if (chi >= 0.0) CHK(spec_queue(x, l, u, g, head, col, stp));
        CHK(fetch(4 * MCo + 9 + NX, 1, 1));
        if (chi >= 0.0) CHK(spec_land(col, chi));
        iw_dirty += h_res[4 * MCo + 8];
      }
      nrpre.valid = false;
      if (NX) {  // formk's new row/column with the pre-walk free set (update_scan_kernel NEWROW)
        const int X0 = 4 * MCo + 9, nold_ = col - 1;
        for (int k = 0; k < 4; ++k) {
          for (int j = 0; j < nold_; ++j) nrpre.t[k][j] = h_res[X0 + k * MCo + j];
          nrpre.t[k][nold_] = h_res[X0 + 4 * MCo + k];
        }
        nrpre.valid = true, nrpre.col = col;
      }
      spec.valid = false;
      tbrk_valid = false;
      pend.on = 1, pend.stp = stp, pend.impl = d_impl ? 1 : 0;  // committed by this call's subspace pass
      rr = h_res[2 * MCo];
      const int nold = col - 1;
      for (int j = 0; j < nold; ++j) {
        scan.p[j] = h_res[2 * MCo + 1 + j];
        scan.p[col + j] = h_res[3 * MCo + 2 + j];
      }
      scan.p[col - 1] = h_res[3 * MCo + 1];
      scan.p[2 * col - 1] = h_res[4 * MCo + 2];
      scan.f1 = h_res[4 * MCo + 3], scan.nbreak = h_res[4 * MCo + 4];
      scan.nunb = h_res[4 * MCo + 5], scan.nunbnz = h_res[4 * MCo + 6];
      scan.bkmin = h_res[4 * MCo + 9 + NX];
      scan.ready = true;
    } else {
      CHK(ensure_d(x));
      lbk::launch_update_pairs<T>(q, n, g, r, d, stp, W(), head, col, itail);
      CHK(fetch(2 * MCo + 1, 0, 0));
      rr = h_res[2 * MCo];
    }
    theta = rr / dr;
    matupd_small(col, iupdat, unfused ? wsy.data() : h_res, unfused ? wss.data() : h_res + MCo,
                 stp == 1.0 ? dtd : stp * stp * dtd, dr);
    info = lbh::formt(m, wt.data(), sy.data(), ss.data(), col, theta);  // :849
    if (info != 0) {
      if (ipr >= 1)
        std::fprintf(rep.out,
                     "\n Nonpositive definiteness in Cholesky factorization in formt;\n   "
                     "refresh the lbfgs memory and restart the iteration.\n");
      refresh(L);
    }
    prelims = linesearch = true;
    seg(2);
    return 0;
  }
#undef MAINLB_VIEW

  int setulb_dev(void *x_, const void *l_, const void *u_, const int32_t *nbd, double *f,
                 void *g_, double factr, double pgtol, char *task, int iprint, char *csave,
                 int32_t *lsave, int32_t *isave_user, double *dsave) override {
    if (!task) return fail(LBFGSB_E_ARG, "setulb: NULL argument");
    if (lbh::str60_eq(task, "START") || entry_mode == 0) {
      if (lbh::str60_eq(task, "START")) t = t_own, r = r_own;
      pp = false, entry_mode = 1, check_ptrs = true;
    } else if (entry_mode != 1) {
      return fail(LBFGSB_E_STATE, "this run was started with lbfgsb_hip_setulb_dev_pp");
    }
    Mainlb L;
    L.x = (T *)x_, L.g = (T *)g_;
    return drive(L, l_, u_, nbd, f, factr, pgtol, task, iprint, csave, lsave, isave_user, dsave);
  }

  // the same with ping-pong iterate buffers (include/lbfgsb_hip.h): x0/x1 and g0/g1 are two pairs of
  // caller buffers; *cur tells which pair this return refers to
  int setulb_dev_pp(void *x0, void *x1, const void *l_, const void *u_, const int32_t *nbd, double *f,
                    void *g0, void *g1, double factr, double pgtol, char *task, int iprint, char *csave,
                    int32_t *lsave, int32_t *isave_user, double *dsave, int32_t *cur) override {
    if (!task) return fail(LBFGSB_E_ARG, "setulb: NULL argument");
    if (!x0 || !x1 || !g0 || !g1 || x0 == x1 || g0 == g1)
      return fail(LBFGSB_E_ARG, "setulb_dev_pp needs two distinct x and two distinct g buffers");
    for (const void *p : {(const void *)x1, (const void *)g1})
      if (((uintptr_t)p & 15) != 0) return fail(LBFGSB_E_ARG, "device pointers must be 16-byte aligned");
    if (lbh::str60_eq(task, "START") || entry_mode == 0) {
      // (entry_mode == 0 without START: a run resumed from lbfgsb_hip_import_state -- the iterate and
      //  its gradient are in x0 / g0, the imported t and r in the context's own buffers)
      if (lbh::str60_eq(task, "START")) t = t_own, r = r_own;
      pp = true, pp_cur = 0, entry_mode = 2, check_ptrs = true;
      xb[0] = (T *)x0, xb[1] = (T *)x1, gb[0] = (T *)g0, gb[1] = (T *)g1;
    } else if (entry_mode != 2) {
      return fail(LBFGSB_E_STATE, "this run was started with lbfgsb_hip_setulb_dev");
    } else if (xb[0] != (T *)x0 || xb[1] != (T *)x1 || gb[0] != (T *)g0 || gb[1] != (T *)g1) {
      return fail(LBFGSB_E_ARG, "setulb_dev_pp: the four buffers must not change during a run");
    }
    Mainlb L;
    L.x = xb[pp_cur], L.g = gb[pp_cur];
    const int rc = drive(L, l_, u_, nbd, f, factr, pgtol, task, iprint, csave, lsave, isave_user, dsave);
    if (cur) *cur = pp_cur;
    return rc;
  }

  int drive(Mainlb &L, const void *l_, const void *u_, const int32_t *nbd, double *f, double factr,
            double pgtol, char *task, int iprint, char *csave, int32_t *lsave, int32_t *isave_user,
            double *dsave) {
    if (!L.x || !L.g || !l_ || !u_ || !nbd || !f || !task || !csave || !lsave || !isave_user || !dsave)
      return fail(LBFGSB_E_ARG, "setulb: NULL argument");   // (never a kernel launched on a null pointer)
    HIPCHK(hipSetDevice(device));
    print_level = iprint;
    quiet = rank != 0;
    L.l = (const T *)l_, L.u = (const T *)u_, L.nbd = nbd, L.f = f;
    L.factr = factr, L.pgtol = pgtol, L.task = task, L.csave = csave, L.lsave = lsave;
    L.isave = isave_user + 21, L.dsave = dsave, L.ipr = quiet ? -1 : iprint;
    for (const void *p : {(const void *)L.x, (const void *)L.l, (const void *)L.u, (const void *)L.g})
      if (((uintptr_t)p & 15) != 0) return fail(LBFGSB_E_ARG, "device pointers must be 16-byte aligned");
    if (lbh::str60_eq(task, "START")) nbd8_src = nullptr;  // (a new run may reuse the buffer)
    if (check_ptrs) {  // START, and the first call of a run resumed from import_state
      check_ptrs = false;
      // a host array handed to this entry by mistake would fault the GPU in the first kernel: every n-vector
      // must be memory the device can address (device, managed, or pinned host memory)
      const void *vec[] = {L.x, L.g, L.l, L.u, nbd, pp ? (const void *)xb[1] : (const void *)L.x,
                           pp ? (const void *)gb[1] : (const void *)L.g};
      for (const void *p : vec) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type == hipMemoryTypeUnregistered ||
            at.devicePointer == nullptr) {
          (void)hipGetLastError();
          check_ptrs = true;  // (the next attempt is looked at again)
          return fail(LBFGSB_E_ARG, "setulb_dev: x, g, l, u, nbd must be device-accessible memory (got a plain host pointer)");
        }
      }
    }
    if (ub_mask && !lbh::str60_eq(task, "START")) {  // other arrays than the ones START looked at
      if (ub_mask & lbk::UB_DICT) {
        // (the code bytes were made from all three arrays: any other array ends the dictionary mode)
        if (L.l != ub_l || L.u != ub_u || nbd != ub_nbd) ub_mask = 0, nbd8_src = nullptr;
      } else {
        if (L.l != ub_l) ub_mask &= ~1;
        if (L.u != ub_u) ub_mask &= ~2;
        if (nbd != ub_nbd) ub_mask &= ~4;
      }
    }
    CHK(ensure_nbd8(nbd));
    Flow flow = NEXT;
#define PHASE(call)               \
  {                               \
    flow = NEXT;                  \
    CHK(call);                    \
    if (flow == DONE) return 0;   \
  }
    if (lbh::str60_eq(task, "START")) {
      PHASE(phase_start(L, flow));
      return 0;
    }
    phase_restore(L);
    PHASE(phase_entry(L, flow));
    if (L.compute_pg) PHASE(phase_first_projgr(L, flow));
    for (;;) {  // main_loop :599
      if (L.prelims) {
        PHASE(phase_cauchy_freev(L, flow));
        if (flow == AGAIN) continue;
        PHASE(phase_subspace(L, flow));
        if (flow == AGAIN) continue;
      }
      if (L.linesearch) {
        do {
          PHASE(phase_linesearch(L, flow));
        } while (flow == REPEAT);
        if (flow == AGAIN) continue;
      }
      PHASE(phase_termination(L, flow));
      PHASE(phase_update(L, flow));
      // (AGAIN and NEXT alike: the next loop trip)
    }
#undef PHASE
  }

  int64_t nfree_g = 0, nenter_g = 0, ileave_g = 0;
  // formk's patch sums fetched together with freev's counts (phase_cauchy_freev)
  struct Eager {
    bool valid = false;
    int upcl = 0, head = 0;
    std::vector<double> P;
  } eager;
  bool eager_on = true;  // (option "eager_patch")
  int64_t neager_served = 0;
  // the same chain queued SPECULATIVELY behind the evaluation of a trial point (phase_entry)
  struct SpecFreev {
    bool valid = false, served = false;
    double cnt[4] = {0, 0, 0, 0}, loc3 = 0.0;
    int parity = 0, upcl = 0, head = 0, col = 0, iter = 0;
    const void *x = nullptr;
    std::vector<double> P;
  } sfv;
  // Off unless option "spec_freev" = 1: measured on BASELINE configs[2] (Rosenbrock n = 1e7, where the free set
  // changes in most iterations) the chain is used in a quarter of the iterations only -- 2.6 -> 2.4 syncs --
  // and its three extra launches per iteration cost more than that saves: 980 -> 955 it/s
  bool spec_freev_on = false;
  bool sfv_hot = false;       // the last freev found the free set changed
  int64_t nsfv_used = 0;
  bool check_ptrs = false;   // this call decides the entry of a run: its pointers have not been looked at yet
  bool index_valid = false;  // a freev has run: wasfree is the membership of Index(1:nfree)
  // How many iwhere entries have changed since the last freev pass (a count where the kernels report
  // one, >= 1 where they do not).  Zero at the point where freev is due means: the free set is what
  // the last freev left -- no variable entered or left (:2012-2035 would find nothing), nfree is
  // unchanged -- and the counting pass and its host sync are skipped (in the steady state of a
  // bounded problem most iterations cross no breakpoint and change no status).
  double iw_dirty = 1.0;
  int64_t nfreev_skipped = 0;

#include "solver_state.inl"     // state exchange, per-kernel doors, communicators
#include "solver_doors.inl"     // routine doors (active, errclb, cauchy, freev, formk, cmprlb, subsm, lnsrlb, matupd)
};

}  // namespace

lbfgsb_hip_ctx *lbfgsb_make_solver(int64_t n_local, int64_t n_global, int64_t row0, int m, int flags,
                                   int device, void *stream, int *rc) {
  auto make = [&](auto *s) -> lbfgsb_hip_ctx * {
    *rc = s->init(n_local, n_global, row0, m, flags, device, stream);
    if (*rc) {
      delete s;
      return nullptr;
    }
    return s;
  };
  if (flags & LBFGSB_F_REAL32) return make(new Solver<float>());
  return make(new Solver<double>());
}
