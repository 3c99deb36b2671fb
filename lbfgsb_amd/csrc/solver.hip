// solver.hip -- host side of the MI355X-native L-BFGS-B inner iteration.
//
// Mirrors the reverse-communication state machine of the reference `mainlb`
// (src/lbfgsb.f90:312-949): same task protocol, same isave/dsave/lsave slots,
// same failure/refresh branches.  Every n-dimensional operation is a kernel
// launch from the k_*.hip files; every 2m x 2m operation is host code from
// host_dense.hpp.  One host thread per context, one HIP stream per context.
//
// One steady-state iteration on a bounded problem with col <= 20 pairs stored (DESIGN.md 4a
// has the why) -- two passes over W:
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
// Fallback with a third pass (cmprlb_wtv: r, W'r, formk's new row): col > 20, few free variables,
// long walks.  Other paths (col = 0, restarts, unconstrained): projgr, cauchy_scan,
// cauchy_finish, formk_gram, update_pairs, lnsrlb_begin/step, pair_commit, xcp_fill,
// subsm_dir/backtrack.
//
// There is no CPU fallback anywhere in this file.
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
  T *xb[2] = {nullptr, nullptr}, *gb[2] = {nullptr, nullptr};
  int pp_cur = 0;         // the pair the last return referred to
  const T *x_lean = nullptr;  // where the first trial point of a lean subspace pass lives (d = x_lean - t)
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
    auto F = [](auto *&p) {
      if (p) (void)hipFree(p);
      p = nullptr;
    };
    F(ws), F(wy), F(zero_buf), F(z), F(r_own), F(d), F(t_own), F(xp), F(tbrk), F(iwhere), F(nbd8), F(index), F(indx2),
        F(scan_tmp), F(wasfree), F(prevfree), F(keys[0]), F(keys[1]), F(idx[0]), F(idx[1]),
        F(sort_tmp), F(d_count), F(d_chg), F(d_msg), F(d_msg2), F(d_msg_all), F(q.d_part), F(q.d_res), F(q.d_gpart),
        F(d_fix), F(pg_buf), F(pg_tmp), F(sp_keys), F(sp_idx), F(sp_count), F(sp_msg), F(sp_msg_all), F(d_res_all);
    F(hx), F(hg), F(hl), F(hu), F(hnbd);
    auto H = [](auto *&p) {
      if (p) (void)hipHostFree(p);
      p = nullptr;
    };
    H(h_count), H(h_msg_all), H(h_msg_loc), H(h_hdr), H(h_res), H(h_fix), H(h_sp_all), H(h_sp_loc), H(h_res_all);
    if (pf_ev) (void)hipEventDestroy(pf_ev);
    pf_ev = nullptr;
    if (order_ev) (void)hipEventDestroy(order_ev);
    order_ev = nullptr;
    for (auto &pair : clk_ev)
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
    ld = ((n + 31) / 32) * 32;
    // streamed-once data: nontemporal loads unless W fits the 256 MiB Infinity Cache
    q.nt = (size_t)2 * ld * m * sizeof(T) > ((size_t)192 << 20);
    const size_t wbytes = (size_t)ld * m * sizeof(T);
    HIPCHK(hipMalloc(&ws, wbytes));
    HIPCHK(hipMalloc(&wy, wbytes));
    HIPCHK(hipMemsetAsync(ws, 0, wbytes, stream));
    HIPCHK(hipMemsetAsync(wy, 0, wbytes, stream));
    HIPCHK(hipMalloc(&zero_buf, 256));  // read by the unroll slots beyond the stored pairs
    HIPCHK(hipMemsetAsync(zero_buf, 0, 256, stream));
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
    res_len = std::max<size_t>(lbk::RES_MAX, E) + 8;
    HIPCHK(hipMalloc(&q.d_part, (size_t)lbk::RES_MAX * lbk::MAX_BLOCKS * sizeof(double)));
    HIPCHK(hipMalloc(&q.d_res, res_len * sizeof(double)));
    HIPCHK(hipMalloc(&q.d_gpart, E * lbk::GRAM_BLOCKS * sizeof(double)));
    HIPCHK(hipHostMalloc(&h_res, res_len * sizeof(double)));
    // cauchy selection scratch (window mode); the full-sort buffers grow on demand
    CHK(ensure_sel(SEL_CAP));
    HIPCHK(hipMalloc(&d_count, sizeof(uint32_t)));
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
    HIPCHK(hipHostMalloc(&h_msg_all, (size_t)nr * msg_len * sizeof(double)));
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
  int fetch(int nsum, int nmin, int nmax) {
    const int k = nsum + nmin + nmax;
    if (q.launch_err != hipSuccess) {  // a kernel launch of this phase failed: name it
      const hipError_t e = q.launch_err;
      q.launch_err = hipSuccess;
      return fail(LBFGSB_E_NOGPU, std::string("kernel launch failed in ") +
                                      (q.launch_err_where ? q.launch_err_where : "?") + ": " +
                                      hipGetErrorString(e));
    }
    if ((size_t)k > res_len) return fail(LBFGSB_E_STATE, "fetch: more partials than the buffer holds");
    if (comm || nranks > 1) ncoll++, coll_bytes += (int64_t)k * 8;
    if (comm) {
      if (g_rccl.AllGather(q.d_res, d_res_all, (size_t)k, ncclDouble, comm, stream) != ncclSuccess)
        return fail(LBFGSB_E_COMM, "ncclAllGather of the partial sums failed");
      HIPCHK(hipMemcpyAsync(h_res_all, d_res_all, (size_t)nranks * k * sizeof(double),
                            hipMemcpyDeviceToHost, stream));
    } else {
      HIPCHK(hipMemcpyAsync(h_res, q.d_res, (size_t)k * sizeof(double), hipMemcpyDeviceToHost,
                            stream));
    }
    {
      const double t0 = now_s();
      HIPCHK(hipStreamSynchronize(stream));
      t_wait += now_s() - t0;
    }
    nsync++;
    if (clock_on) clk_collect();
    if (comm) {
      for (int j = 0; j < k; ++j) {
        double v = h_res_all[j];
        for (int rk = 1; rk < nranks; ++rk) {
          const double w = h_res_all[(size_t)rk * k + j];
          v = j < nsum ? v + w : (j < nsum + nmin ? std::fmin(v, w) : std::fmax(v, w));
        }
        h_res[j] = v;
      }
    } else if (nranks > 1) {
      if (!cb_ar) return fail(LBFGSB_E_COMM, "multi-rank context without a reducer");
      if (cb_ar(cb_user, h_res, nsum, nmin, nmax) != 0)
        return fail(LBFGSB_E_COMM, "host all-reduce callback failed");
    }
    return 0;
  }

  lbk::WStore<T> W() const { return lbk::WStore<T>{ws, wy, ld, m, zero_buf}; }

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
    const double *raw = nullptr;  // single rank, col = 0: the chunk itself is in order (records of
                                  // 4 doubles); M is then only sized, not filled
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
      lbk::launch_cauchy_window_fly<T>(q, n, row0, x, l, u, nbd8, g, iwhere, lo_t, lo_i, hi, keys[0],
                                       idx[0], SEL_CAP, d_count);
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
      CHK(exchange(count));
    }
    pf_valid = false;
    const bool rawmode = single && col == 0 && print_level < 100 && !debug_walk;
    pv.raw = nullptr;
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
      const size_t at = pv.M.size();
      pv.M.resize(at + lr);
      if (rawmode) {
        pv.raw = base + 2;
      } else {
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
    pv.safe_end = pv.M.size();
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
    if (fix_overflow) {  // long walk: the cursor-based kernel (it writes z on the way)
      CHK(ensure_tbrk());
      lbk::launch_cauchy_finish<T>(q, n, row0, (const T *)cx, (const T *)cl, (const T *)cu,
                                   (const T *)cg, tbrk, iwhere, z, tsum, last_t, last_i);
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
    double p[2 * lbk::MAXM];
    double f1 = 0, nbreak = 0, nunb = 0, nunbnz = 0, bkmin = 0;
  } scan;

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
    if (!scan.ready) {
      lbk::launch_cauchy_scan<T>(q, n, x, l, u, nbd, g, iwhere, tbrk, W(), head, col);
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
            pv.taken[pv.raw ? 0 : M[pos].rank]++;
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
          if (pv.have && (pv.mpos < pv.M.size() || pv.more_anywhere)) {
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
                              print_level < 99;
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

  // ==================================================================== formk
  // WN1 from scratch: one masked Gram pass over W (any col; also the fallback when too many
  // variables changed status for the sparse patches)
  int formk_scratch(int col, int head) {
    CHK(commit_pending((const T *)cg, col, head));
    lbk::launch_formk_gram<T>(q, n, W(), head, col, iwhere);
    const int E = 2 * col * col + col;
    CHK(fetch(E, 0, 0));
    lbh::Mat WN1{snd.data(), 2 * m};
    const int tri = col * (col + 1) / 2;
    for (int i = 0; i < col; ++i)
      for (int j = 0; j <= i; ++j) {
        WN1(i, j) = h_res[i * (i + 1) / 2 + j];                  // Y'ZZ'Y
        WN1(m + i, m + j) = h_res[tri + i * (i + 1) / 2 + j];    // S'AA'S
      }
    for (int i = 0; i < col; ++i)
      for (int j = 0; j < col; ++j) WN1(m + i, j) = h_res[2 * tri + i * col + j];  // L_a + R_z
    return 0;
  }

  // WN1 kept incrementally exactly as the reference does (:1735-1851): shift, new row and
  // column from `nr` (the four sum vectors that rode along in the cmprlb_wtv pass), and
  // patches for the variables that entered / left the free set (sparse signed Gram).
  int formk_incremental(int col, int head, bool updatd, int iupdat, const double *nr, int MCnr) {
    const int m2 = 2 * m;
    lbh::Mat WN1{snd.data(), m2};
    const int upcl = updatd ? col - 1 : col;
    const int64_t nchg = nenter_g + (nglob + 1 - ileave_g);
    bool patched = false;
    std::vector<double> P;
    if (nchg > 0 && upcl > 0) {
      if (nchg > (int64_t)CHG_CAP) return formk_scratch(col, head);  // whole Gram is cheaper
      // (the list was appended with an atomic counter: put it in ascending order first, so that the
      //  patch sums -- and with them WN1, the subspace step, the whole trajectory -- are
      //  reproducible bit for bit; idx[1] is free here: the walk is over)
      const uint32_t nl = std::min<uint32_t>(chg_local, CHG_CAP);
      const uint32_t *lst = lbk::launch_sort_u32(q, sort_tmp, sort_tmp_bytes, d_chg, idx[1], nl);
      lbk::launch_formk_patch<T>(q, lst, nl, W(), head, upcl);
      const int E = 2 * upcl * upcl + upcl;
      CHK(fetch(E, 0, 0));
      P.assign(h_res, h_res + E);
      patched = true;
    }
    if (updatd) {
      if (iupdat > m) {  // shift old part of WN1 (:1736-1744)
        for (int jy = 0; jy < m - 1; ++jy) {
          const int js = m + jy;
          for (int i = 0; i < m - 1 - jy; ++i) {
            WN1(jy + i, jy) = WN1(jy + 1 + i, jy + 1);
            WN1(js + i, js) = WN1(js + 1 + i, js + 1);
          }
          for (int i = 0; i < m - 1; ++i) WN1(m + i, jy) = WN1(m + 1 + i, jy + 1);
        }
      }
      const int nw = col - 1;  // new pair = logical column col-1 (:1746-1793)
      for (int jy = 0; jy < col; ++jy) {
        WN1(nw, jy) = nr[0 * MCnr + jy];          // Y'ZZ'Y row
        WN1(m + nw, m + jy) = nr[1 * MCnr + jy];  // S'AA'S row
        WN1(m + nw, jy) = nr[2 * MCnr + jy];      // L_a row
      }
      for (int i = 0; i < col; ++i) WN1(m + i, nw) = nr[3 * MCnr + i];  // R_z column
    }
    if (patched) {  // :1801-1851 (P = sums over entering rows - sums over leaving rows)
      const int tri = upcl * (upcl + 1) / 2;
      for (int iy = 0; iy < upcl; ++iy)
        for (int jy = 0; jy <= iy; ++jy) {
          WN1(iy, jy) = WN1(iy, jy) + P[iy * (iy + 1) / 2 + jy];
          WN1(m + iy, m + jy) = WN1(m + iy, m + jy) - P[tri + iy * (iy + 1) / 2 + jy];
        }
      for (int is = 0; is < upcl; ++is)
        for (int jy = 0; jy < upcl; ++jy) {
          const double psy = P[2 * tri + is * upcl + jy];
          if (is <= jy)
            WN1(m + is, jy) = WN1(m + is, jy) + psy;
          else
            WN1(m + is, jy) = WN1(m + is, jy) - psy;
        }
    }
    return 0;
  }

  // upper triangle of WN from WN1 and the two Cholesky factorisations (:1856-1906)
  void formk_factor(int col, double theta, int &info) {
    const int m2 = 2 * m;
    lbh::Mat WN{wn.data(), m2}, WN1{snd.data(), m2}, SY{sy.data(), m};
    for (int iy = 0; iy < col; ++iy) {
      const int is = col + iy, is1 = m + iy;
      for (int jy = 0; jy <= iy; ++jy) {
        const int js = col + jy, js1 = m + jy;
        WN(jy, iy) = WN1(iy, jy) / theta;
        WN(js, is) = WN1(is1, js1) * theta;
      }
      for (int jy = 0; jy < iy; ++jy) WN(jy, is) = -WN1(is1, jy);
      for (int jy = iy; jy < col; ++jy) WN(jy, is) = WN1(is1, jy);
      WN(iy, iy) = WN(iy, iy) + SY(iy, iy);
    }
    if (lbh::dpofa(WN, col) != 0) {  // :1880-1884
      info = -1;
      return;
    }
    const int col2 = 2 * col;
    for (int js = col; js < col2; ++js) (void)lbh::dtrsl(WN, col, &WN(0, js), 11);
    for (int is = col; is < col2; ++is)
      for (int js = is; js < col2; ++js)
        WN(is, js) = WN(is, js) + lbh::dot_seq(col, &WN(0, is), &WN(0, js));
    lbh::Mat WN22{&WN(col, col), m2};
    if (lbh::dpofa(WN22, col) != 0) {  // :1902-1906
      info = -2;
      return;
    }
    info = 0;
  }

  int formk(int col, int head, double theta, int &info) {
    CHK(formk_scratch(col, head));
    formk_factor(col, theta, info);
    return 0;
  }

  // ========================================================== cmprlb + subsm
  // coefficients of cmprlb: wa(1:2m) = M c (bmv, :1569) -> a1_j, a2_j = theta * (.) (:1576-1577)
  // (kept in cm_cf / cm_plain: subsm_update_kernel recomputes r from them)
  lbk::Coef cm_cf;
  bool cm_plain = false;
  bool cmprlb_coef(int col, double theta, bool cnstnd, lbk::Coef &cf, bool &plain) {
    std::memset(&cf, 0, sizeof cf);
    plain = !cnstnd && col > 0;
    if (!plain) {
      if (lbh::bmv(m, sy.data(), wt.data(), col, &wa8m[2 * m], &wa8m[0]) != 0) return false;
      for (int j = 0; j < col; ++j) {
        cf.a[j] = wa8m[j];
        cf.a[lbk::MAXM + j] = theta * wa8m[col + j];
      }
    }
    cm_cf = cf, cm_plain = plain;
    return true;
  }

  // W'Z r without a pass over W (cmprlb :1565-1583 folded into subsm :2742-2754).  On the free
  // rows the Cauchy point is x + tsum d with d = -g, so
  //     r = (1 - theta tsum) d + W (M c)   on the free rows Z,   and
  //     W'Z r = (1 - theta tsum) W'Z d + (W'ZZ'W) (M c).
  // W'Z d is the p the walk ends with (it carries W'd over the variables that still move,
  // :1300-1304, :1463-1470); W'ZZ'W is in WN1 and in matupd's matrices:  Y'ZZ'Y = WN1(1:col,1:col),
  // S'ZZ'S = S'S - S'AA'S = Ss - WN1(m+1:,m+1:),  S'ZZ'Y = R_z above the diagonal (WN1), Sy - L_a
  // below it (:1756-1793).  Equal to the sums over the rows up to reassociation -- and to the
  // rounding of z - x, which the row form carries at 1 ulp of x per row: the caller uses this
  // form only when neither the free set nor p is a small remainder of something much larger.
  // ... which is the case while every stored s_i keeps at least 1e-5 of its squared norm on the free
  // rows (variables that sit at a bound do not move: their part of s is zero unless they have
  // just arrived, so a small free SET alone does not make the free PART small)
  // The same kind of difference gives sum_free s_i y_j below the diagonal: Sy(i,j) - L_a(i,j)
  // (total minus active).  Entry by entry that difference may be small against its operands without
  // harm -- what must not drown is its contribution to W'Z r, which is measured against the free
  // norms |Z's_i| |Z'y_j| (Cauchy-Schwarz bounds the exact value by them): the rounding error of the
  // difference, ~eps (|Sy| + |L_a|), has to stay below 1e-5 of that scale.
  bool closed_form_safe(int col) const {
    const double *WN1 = snd.data(), *SS = ss.data(), *SY = sy.data();
    const int m2 = 2 * m;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int i = 0; i < col; ++i) {
      const double tot = SS[(size_t)i + (size_t)i * m];
      const double act = WN1[(size_t)(m + i) + (size_t)(m + i) * m2];
      if (!(tot - act >= 1.0e-5 * tot)) return false;
    }
    for (int i = 1; i < col; ++i) {
      const double ssf = SS[(size_t)i + (size_t)i * m] - WN1[(size_t)(m + i) + (size_t)(m + i) * m2];
      for (int j = 0; j < i; ++j) {
        const double yyf = WN1[(size_t)j + (size_t)j * m2];
        const double tot = SY[(size_t)i + (size_t)j * m], act = WN1[(size_t)(m + i) + (size_t)j * m2];
        const double scale = std::sqrt(std::fabs(ssf) * std::fabs(yyf));
        if (!(eps * (std::fabs(tot) + std::fabs(act)) <= 1.0e-5 * scale)) return false;
      }
    }
    return true;
  }
  void subspace_closed_form(int col, double theta, double *wv) {
    const int m2 = 2 * m;
    lbh::Mat WN1{snd.data(), m2}, SY{sy.data(), m}, SS{ss.data(), m};
    const double k1 = 1.0 - theta * gcp.tsum;
    auto YYf = [&](int i, int j) { return i >= j ? WN1(i, j) : WN1(j, i); };
    auto SSf = [&](int i, int j) {
      const double tot = i <= j ? SS(i, j) : SS(j, i);
      const double act = i >= j ? WN1(m + i, m + j) : WN1(m + j, m + i);
      return tot - act;
    };
    auto SYf = [&](int is, int jy) {  // sum_free s_is y_jy
      return is <= jy ? WN1(m + is, jy) : SY(is, jy) - WN1(m + is, jy);
    };
    const double *a1 = cm_cf.a, *a2 = cm_cf.a + lbk::MAXM;  // (M c)_j, theta (M c)_{col+j}
    for (int i = 0; i < col; ++i) {
      double ay = k1 * p_fin[i], as = k1 * (p_fin[col + i] / theta);
      for (int j = 0; j < col; ++j) {
        ay = ay + YYf(i, j) * a1[j] + SYf(j, i) * a2[j];
        as = as + SYf(i, j) * a1[j] + SSf(i, j) * a2[j];
      }
      wv[i] = ay;
      wv[col + i] = theta * as;
    }
  }

  // do_formk: formk is pending for this iteration and col <= 10: its new row/column sums ride
  // along in the cmprlb_wtv pass and the status changes are patched sparsely.
  // closed: no cmprlb pass at all -- new row from the update pass (nrpre, corrected by the
  // walk), W'Z r in closed form.
  int subspace(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, double theta,
               int col, int head, bool cnstnd, int &iword, int &info, bool do_formk, bool updatd,
               int iupdat, const double *pre, bool closed = false) {
    // cmprlb :1548-1586 (+ W'r of subsm).  `pre` != nullptr: the pass was already launched
    // together with freev's counts (one fetch for both) and its sums are in pre[].
    const int MC = lbk::maxc_for(col);
    const bool newrow = do_formk && updatd;
    const double *res = pre;
    const int ipr = quiet ? -1 : print_level;
    if (closed) {
      lbk::Coef cf;
      bool plain;
      if (!cmprlb_coef(col, theta, cnstnd, cf, plain)) {
        info = -8;
        return 0;
      }
      nclosed++;
    } else if (!pre) {
      lbk::Coef cf;
      bool plain;
      if (!cmprlb_coef(col, theta, cnstnd, cf, plain)) {
        // (the reference would run formk first, :663; either failure refreshes the memory,
        //  after which WN1 is rebuilt from new rows only)
        info = -8;
        return 0;
      }
      CHK(ensure_d(x));
      clk_begin(0);
      lbk::launch_cmprlb_wtv<T>(q, n, x, g, gcp.tsum, iwhere, W(), head, col, theta, cf,
                                newrow ? 1 : 0, r, d, pend);
      clk_end(0);
      CHK(fetch((newrow ? 6 : 2) * MC, 0, 0));
      res = h_res;
    }
    double *wv = &wa8m[0];
    if (!closed) {
      nthreepass++;
      for (int i = 0; i < col; ++i) {
        wv[i] = res[i];
        wv[col + i] = theta * res[MC + i];
      }
    }
    if (do_formk) {
      double nr[4 * lbk::MAXM];
      if (closed) {
        if (newrow) {
          for (int j = 0; j < col; ++j) {
            nr[0 * MC + j] = nrpre.t[0][j] - nrc[0][j];
            nr[1 * MC + j] = nrpre.t[1][j] + nrc[1][j];
            nr[2 * MC + j] = nrpre.t[2][j] + nrc[2][j];
            nr[3 * MC + j] = nrpre.t[3][j] - nrc[3][j];
          }
        }
      } else if (newrow) {
        std::memcpy(nr, res + 2 * MC, sizeof(double) * 4 * MC);
      }
      CHK(formk_incremental(col, head, updatd, iupdat, nr, MC));
      formk_factor(col, theta, info);
      if (info != 0) return 0;
    }
    if (closed && !closed_form_safe(col)) {
      // the free part of some s_i is a tiny remainder of the whole column: S'ZZ'S = S'S - S'AA'S
      // would lose it to cancellation.  W'Z r from a pass over W after all (WN1 is complete: no
      // new-row sums)
      CHK(ensure_d(x));
      clk_begin(0);
      lbk::launch_cmprlb_wtv<T>(q, n, x, g, gcp.tsum, iwhere, W(), head, col, theta, cm_cf, 0, r, d, pend);
      clk_end(0);
      CHK(fetch(2 * MC, 0, 0));
      for (int i = 0; i < col; ++i) {
        wv[i] = h_res[i];
        wv[col + i] = theta * h_res[MC + i];
      }
      closed = false;
      nclosed--, nthreepass++;
    }
    if (closed) subspace_closed_form(col, theta, wv);
    if (ipr >= 99) std::fprintf(rep.out, "\n----------------SUBSM entered-----------------\n\n");  // :2738
    lbh::Mat WN{wn.data(), 2 * m};
    const int col2 = 2 * col;
    info = lbh::dtrsl(WN, col2, wv, 11);
    if (info != 0) return 0;
    for (int i = 0; i < col; ++i) wv[i] = -wv[i];
    info = lbh::dtrsl(WN, col2, wv, 1);
    if (info != 0) return 0;
    lbk::Coef cw;
    std::memset(&cw, 0, sizeof cw);
    for (int j = 0; j < col; ++j) {
      cw.a[j] = wv[j];
      cw.a[lbk::MAXM + j] = wv[col + j];
    }
    // d, t, r get their line-search values in the same pass (see subsm_update_kernel); xp = xcp
    // (:2787) is written out only for state export -- and below if the backtracking branch runs
    if (flags & LBFGSB_F_MIRROR_INDEX) CHK(write_xcp(xp, x, l, u, g));
    // lean: the first trial step is 1 and x = z is stored by the pass, so neither z nor d = x - t
    // is written (5 store streams instead of 7); they stay implicit until ensure_d()
    const bool lean = lean_on && ls_unit_step && (cnstnd || two_pass) && !(flags & LBFGSB_F_MIRROR_INDEX);
    clk_begin(2);
    // (ping-pong buffers: no t = x, r = g copies -- the roles change below -- and the trial point
    //  goes to the other x buffer, which is where the pending pair's t is read from, row by row)
    lbk::launch_subsm_update<T>(q, n, gcp.tsum, lean ? (T *)nullptr : z, r, pp ? (T *)nullptr : r, l, u, nbd8,
                                iwhere, x, g, W(), head, col, theta, cm_cf, cw, lean ? (T *)nullptr : d,
                                pp ? (T *)nullptr : t, ls_unit_step ? xmut : nullptr, ls_do_stpmx ? 1 : 0,
                                pend, d_src());
    clk_end(2);
    pend.on = 0, pend.impl = 0;  // the pass stored the pair into its W slot
    d_impl = z_in_x = lean;
    z_valid = !lean;
    if (lean) x_lean = xmut;
    if (pp) t = const_cast<T *>(x), r = const_cast<T *>(g);  // t = x, r = g (:2235-2236) as a change of roles
    CHK(fetch(3, 1, 0));
    iword = h_res[0] > 0.0 ? 1 : 0;
    const double dd_p = h_res[1];
    ls.ready = true;
    ls.x_is_z = ls_unit_step;
    ls.gd = dd_p;
    ls.dtd = h_res[2];
    ls.stpmx = h_res[3];
    if (iword == 0 || dd_p <= 0.0) {  // :2820, :2828
      if (ipr >= 99) std::fprintf(rep.out, "\n----------------exit SUBSM --------------------\n\n");  // :2883
      return 0;
    }
    ls.ready = false;  // z changes below: lnsrlb_begin redoes d, t, r
    d_impl = z_in_x = false;  // (and the backtracking kernel writes all of z)
    if (ls.x_is_z) {   // ... from the iterate itself, which the pass above saved in t
      // (ping-pong buffers: the trial point went to the other buffer, x still is the iterate)
      if (!pp) HIPCHK(hipMemcpyAsync(xmut, t, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
      ls.x_is_z = false;
    }
    if (rep.out && !quiet && print_level >= 0) {
      std::fprintf(rep.out, " Positive dir derivative in projection \n");
      std::fprintf(rep.out, " Using the backtracking step \n");
    }
    // xp = xcp and the Newton direction as vectors (the direction goes to tbrk, which the
    // cursor-based cauchy_finish_kernel has read by then)
    if (!(flags & LBFGSB_F_MIRROR_INDEX)) CHK(write_xcp(xp, x, l, u, g));
    lbk::launch_subsm_dir<T>(q, n, xp, iwhere, x, g, W(), head, col, theta, cm_cf, cw, tbrk);
    tbrk_valid = false;
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

  // line-search set-up values when they were produced by the subsm pass
  struct LsOut {
    bool ready = false;
    bool x_is_z = false;  // the pass already stored the first trial point x = z
    double gd = 0, dtd = 0, stpmx = 0;
  } ls;
  bool ls_do_stpmx = false;
  bool ls_unit_step = false;  // the first trial step of this iteration's line search is 1
  T *xmut = nullptr;          // the caller's x of this call
  // sums of a cmprlb_wtv pass that was launched together with freev's counts
  double pre_res[6 * lbk::MAXM];
  bool pre_valid = false;
  // ---- two-pass iteration (col <= 10): formk's new row rides in the update pass with the
  //      pre-walk free set, the walk corrects it for the rows it fixes, and W'Z r follows in
  //      closed form from the walk's p and WN1 (subspace_closed_form) -- no cmprlb pass ----
  bool two_pass = true;  // (option "two_pass")
  // (col <= 20: beyond that the update pass has no registers for the 4 col + 4 extra sums;
  //  option "two_pass_maxcol" lowers the limit, for measurements)
  int two_pass_maxcol = 20;
  bool exact_always = false;  // (option "exact_always": every walk in the reference's heap order)

  // lbfgsb_hip_set_option: measurement / test switches of THIS context (include/lbfgsb_hip.h)
  int set_option(const char *name, double v) override {
    const std::string k = name ? name : "";
    const auto flag = [&](bool &dst) -> int {
      if (v != 0.0 && v != 1.0) return fail(LBFGSB_E_ARG, "set_option: " + k + " takes 0 or 1");
      dst = v != 0.0;
      return 0;
    };
    const auto in_range = [&](int lo, int hi, int &dst) -> int {
      if (!(v >= lo && v <= hi) || v != std::floor(v))
        return fail(LBFGSB_E_ARG, "set_option: " + k + " out of range");
      dst = (int)v;
      return 0;
    };
    if (k == "two_pass") return flag(two_pass);
    if (k == "two_pass_maxcol") return in_range(0, 20, two_pass_maxcol);
    if (k == "lean") return flag(lean_on);
    if (k == "spec_capture") return flag(spec_on);
    if (k == "exact_always") return flag(exact_always);
    if (k == "nt") return flag(q.nt);
    if (k == "pg_min") {
      if (!(v >= 0.0)) return fail(LBFGSB_E_ARG, "set_option: pg_min must be >= 0");
      PG_MIN = v;
      return 0;
    }
    if (k == "wgrid") return in_range(1, lbk::MAX_BLOCKS - 1, q.tune.wgrid);
    if (k == "pipe") return in_range(-1, 1, q.tune.pipe);
    if (k == "pair") return in_range(0, 2, q.tune.pair);
    if (k == "gram_rows") return in_range(0, 1, q.tune.gram_rows);
    return fail(LBFGSB_E_ARG, "set_option: unknown option '" + k + "'");
  }
  // update_scan_kernel's NEWROW flag for the pass that forms pair number `colnew`
  int nr_flag(int colnew) const { return two_pass && colnew <= two_pass_maxcol ? 1 : 0; }
  struct NewRow {
    bool valid = false;
    int col = 0;
    double t[4][lbk::MAXM];  // logical columns 0..col-1: Y'ZZ'Y row, S'AA'S row, L_a row, R_z column
  } nrpre;
  double nrc[4][lbk::MAXM];  // what the walk's fixed rows take from / add to them
  double p_fin[2 * lbk::MAXM], p_ini_max = 0.0;
  bool closed_ok = false;    // this call's cauchy left everything the closed form needs
  int64_t nclosed = 0, nthreepass = 0;

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
  };
  // what a phase tells the driver loop: go on with the next phase | start the loop trip again
  // (memory refreshed, update skipped) | the call is over
  enum Flow { NEXT, AGAIN, DONE };
  static int again(Flow &fl) { return fl = AGAIN, 0; }
  static int done(Flow &fl) { return fl = DONE, 0; }
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
    pend.on = 0, pend.impl = 0;  // the memory is dropped, an uncommitted pair with it
  }

  // x = t, g = r (:568-569, :736-737).  Ping-pong buffers: the previous iterate still sits in its
  // own pair, which simply becomes the pair the caller is pointed at again.
  int restore_iterate(Mainlb &L) {
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
    spcand.valid = false, last_tsum = 0.0, last_dtm0 = 0.0, iter_seen = 0, spec_factor = 2.0;
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
    CHK(fetch(0, 0, 2));
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
    if (lbh::str60_pre(task, "FG_LN")) {
      compute_pg = false, prelims = false;
      spec.valid = false;
      spcand.valid = false;
      // First trial of a line search on a bounded problem: it is accepted far more often than
      // not, so evaluate it with the pass that matupd + the next cauchy scan would run anyway
      // (read-only with the pair pending); g'd and |proj g| are two of its sums.  Contexts
      // that mirror the reference's arrays at every return keep iwhere untouched until the
      // update is real (iwhere_update_kernel at the NEW_X entry).
      // Unconstrained problems (two_pass): the same pass -- every row is free, its p = W'd is
      // W'Z r itself (r = -g, c = 0), and the new pair needs no copy pass of its own.
      if ((cnstnd || two_pass) && ifun == 1) {
        const int store_iw = (flags & LBFGSB_F_MIRROR_INDEX) ? 0 : 1;
        int c2, h2, it2;  // matupd's pointer update (:2303-2309), as if this trial is accepted
        if (iupdat + 1 <= m) {
          c2 = iupdat + 1, h2 = head, it2 = (head + iupdat - 1) % m + 1;
        } else {
          c2 = col, it2 = itail % m + 1, h2 = head % m + 1;
        }
        const int MCo = lbk::maxc_for(c2 - 1);
        const int NX = lbk::update_scan_extra(c2 - 1, nr_flag(c2));
        clk_begin(1);
        q.res_off = fo;
        // (the MC = 20 instantiation with the new-row sums has no registers for the hand-over)
        const double chi = (nr_flag(c2) && lbk::maxc_for(c2 - 1) > 10) ? -1.0 : spec_hi(cnstnd);
        lbk::launch_update_scan<T>(q, n, x, l, u, nbd8, g, r, d_src(), d_impl ? 1 : 0, stp, iwhere,
                                   (T *)nullptr, W(), h2, c2, it2, 0, store_iw, nr_flag(c2), chi,
                                   sp_keys, sp_idx, SPEC_CAP, sp_count);
        q.res_off = 0;
        clk_end(1);
        spcand.valid = false;
        if (chi >= 0.0) CHK(spec_queue(x, l, u, g, h2, c2, stp));
        CHK(fetch(fo + 4 * MCo + 9 + NX, 1, 1));
        if (chi >= 0.0) CHK(spec_land(c2, chi));
        if (fo) *f = f_scale * h_res[0];
        const double *R = h_res + fo;
        gd = R[4 * MCo + 7];
        spec_sbgnrm = R[4 * MCo + 10 + NX];
        std::memcpy(spec.res, R, sizeof(double) * (4 * MCo + 11 + NX));
        spec.valid = true;  // dropped below unless dcsrch accepts this point
        spec.x = x, spec.g = g, spec.stp = stp, spec.head = h2, spec.col = c2, spec.itail = it2;
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
    } else if (lbh::str60_pre(task, "NEW_X")) {
      compute_pg = false, prelims = false, linesearch = false;
    } else if (!lbh::str60_pre(task, "FG_ST")) {
      if (lbh::str60_pre(task, "STOP")) {
        if (std::strncmp(task + 6, "CPU", 3) == 0) {  // :566-571
          CHK(restore_iterate(L));
          HIPCHK(hipStreamSynchronize(stream));
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
      std::memset(nrc, 0, sizeof nrc);
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
      if (prevfree)
        HIPCHK(hipMemcpyAsync(prevfree, wasfree, (size_t)n, hipMemcpyDeviceToDevice, stream));
      const bool track = iter > 0 && cnstnd;  // freev looks for entering/leaving rows (:2012)
      lbk::launch_freev_count(q, n, iwhere, wasfree, track ? d_chg : nullptr, CHG_CAP, d_count);
      index_valid = true;
      if (track)
        HIPCHK(hipMemcpyAsync(h_count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      // the cmprlb pass does not depend on freev's counts: launch it now and fetch both
      // sets of sums with ONE host sync (it is wasted only if no variable is free)
      pre_valid = false;
      int npre = 0;
      // (two-pass iteration: no cmprlb pass if the closed form applies -- decided for good
      //  once nfree is known, below)
      const bool closed_cand = two_pass && closed_ok && col > 0 && col <= two_pass_maxcol &&
                               (!updatd || (nrpre.valid && nrpre.col == col));
      if (col > 0 && !closed_cand) {
        lbk::Coef cf;
        bool plain;
        if (cmprlb_coef(col, theta, cnstnd, cf, plain)) {
          const bool newrow = updatd && col <= 20;  // updatd implies wrk
          CHK(ensure_d(x));
          q.res_off = 3;
          clk_begin(0);
          lbk::launch_cmprlb_wtv<T>(q, n, x, g, gcp.tsum, iwhere, W(), head, col, theta, cf,
                                    newrow ? 1 : 0, r, d, pend);
          clk_end(0);
          q.res_off = 0;
          npre = (newrow ? 6 : 2) * lbk::maxc_for(col);
        }
      }
      CHK(fetch(3 + npre, 0, 0));
      if (npre) {
        std::memcpy(pre_res, h_res + 3, sizeof(double) * npre);
        pre_valid = true;
      }
      chg_local = track ? *h_count : 0;
      cachyt += now_s() - cpu1;
      nintol += nseg;
      nfree_g = (int64_t)h_res[0];
      if (iter > 0 && cnstnd) {
        nenter_g = (int64_t)h_res[1];
        ileave_g = nglob + 1 - (int64_t)h_res[2];
      } else {
        nenter_g = 0;
        ileave_g = nglob + 1;
      }
      wrk = (ileave_g < nglob + 1) || (nenter_g > 0) || updatd;
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
      const bool incr = wrk && col <= 20;  // incremental WN1, fused into the cmprlb pass
      if (wrk && !incr) CHK(formk(col, head, theta, info));
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
      CHK(subspace(x, l, u, nbd, g, theta, col, head, cnstnd, iword, info, incr, updatd, iupdat,
                   pre_valid ? pre_res : nullptr, closed));
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
    const bool setup_call = !lbh::str60_pre(task, "FG_LN");  // first call of this iteration's line search
    if (setup_call) {
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
        if (!(ls.x_is_z && ifun == 1 && stp == 1.0)) {  // else x = z is already in place
          CHK(ensure_d(setup_call ? xmut : x));  // (it still holds the rejected first trial point z)
          // (the set-up call writes this iteration's first trial point: xmut -- the caller's x, or the
          //  other buffer of a ping-pong pair; later calls are entered with x = the trial buffer)
          lbk::launch_lnsrlb_step<T>(q, n, setup_call ? xmut : x, z, d, t, stp);
        }
        ls.x_is_z = false;
        spec.valid = false;  // the trial point was not accepted
        spcand.valid = false;
      } else {
        lbh::str60_set(task, "NEW_X");
      }
    } else {
      spec.valid = false;
      spcand.valid = false;
    }

    if (info != 0 || iback >= 20) {  // :734-769
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
      save_locals(L);
      if (!(flags & LBFGSB_F_NO_RETURN_SYNC)) {
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
    const int MCo = lbk::maxc_for(col - 1);
    double rr;
    if (cnstnd || two_pass) {
      // the next loop trip starts with cauchy: do its n-loop in the same pass over W --
      // unless that pass already ran as the evaluation of the accepted trial point
      const bool reuse = spec.valid && spec.x == x && spec.g == g && spec.stp == stp &&
                         spec.head == head && spec.col == col && spec.itail == itail;
      const int NX = lbk::update_scan_extra(col - 1, nr_flag(col));
      if (reuse) {
        std::memcpy(h_res, spec.res, sizeof(double) * (4 * MCo + 11 + NX));
        if ((flags & LBFGSB_F_MIRROR_INDEX) && h_res[4 * MCo + 8] > 0.0)
          lbk::launch_iwhere_update<T>(q, n, x, l, u, nbd, g, iwhere);  // the pass held it back
      } else {
        clk_begin(1);
        const double chi = (nr_flag(col) && lbk::maxc_for(col - 1) > 10) ? -1.0 : spec_hi(cnstnd);
        lbk::launch_update_scan<T>(q, n, x, l, u, nbd8, g, r, d_src(), d_impl ? 1 : 0, stp, iwhere,
                                   (T *)nullptr, W(), head, col, itail, 0, 1, nr_flag(col), chi,
                                   sp_keys, sp_idx, SPEC_CAP, sp_count);
        clk_end(1);
        spcand.valid = false;
        if (chi >= 0.0) CHK(spec_queue(x, l, u, g, head, col, stp));
        CHK(fetch(4 * MCo + 9 + NX, 1, 1));
        if (chi >= 0.0) CHK(spec_land(col, chi));
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
    lbh::Mat SY{sy.data(), m}, SS{ss.data(), m};
    if (iupdat > m) {  // :2324-2330
      for (int j = 0; j < col - 1; ++j) {
        for (int i = 0; i <= j; ++i) SS(i, j) = SS(i + 1, j + 1);
        for (int i = j; i < col - 1; ++i) SY(i, j) = SY(i + 1, j + 1);
      }
    }
    for (int j = 0; j < col - 1; ++j) {
      SY(col - 1, j) = h_res[j];
      SS(j, col - 1) = h_res[MCo + j];
    }
    SS(col - 1, col - 1) = stp == 1.0 ? dtd : stp * stp * dtd;
    SY(col - 1, col - 1) = dr;
    info = lbh::formt(m, wt.data(), sy.data(), ss.data(), col, theta);  // :849
    if (info != 0) {
      if (ipr >= 1)
        std::fprintf(rep.out,
                     "\n Nonpositive definiteness in Cholesky factorization in formt;\n   "
                     "refresh the lbfgs memory and restart the iteration.\n");
      refresh(L);
    }
    prelims = linesearch = true;
    return 0;
  }
#undef MAINLB_VIEW

  int setulb_dev(void *x_, const void *l_, const void *u_, const int32_t *nbd, double *f,
                 void *g_, double factr, double pgtol, char *task, int iprint, char *csave,
                 int32_t *lsave, int32_t *isave_user, double *dsave) override {
    if (lbh::str60_eq(task, "START")) {
      pp = false;
      t = t_own, r = r_own;
    } else if (pp) {
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
    if (!x0 || !x1 || !g0 || !g1 || x0 == x1 || g0 == g1)
      return fail(LBFGSB_E_ARG, "setulb_dev_pp needs two distinct x and two distinct g buffers");
    for (const void *p : {(const void *)x1, (const void *)g1})
      if (((uintptr_t)p & 15) != 0) return fail(LBFGSB_E_ARG, "device pointers must be 16-byte aligned");
    if (lbh::str60_eq(task, "START")) {
      pp = true, pp_cur = 0;
      xb[0] = (T *)x0, xb[1] = (T *)x1, gb[0] = (T *)g0, gb[1] = (T *)g1;
      t = t_own, r = r_own;
    } else if (!pp) {
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
    HIPCHK(hipSetDevice(device));
    print_level = iprint;
    quiet = rank != 0;
    L.l = (const T *)l_, L.u = (const T *)u_, L.nbd = nbd, L.f = f;
    L.factr = factr, L.pgtol = pgtol, L.task = task, L.csave = csave, L.lsave = lsave;
    L.isave = isave_user + 21, L.dsave = dsave, L.ipr = quiet ? -1 : iprint;
    for (const void *p : {(const void *)L.x, (const void *)L.l, (const void *)L.u, (const void *)L.g})
      if (((uintptr_t)p & 15) != 0) return fail(LBFGSB_E_ARG, "device pointers must be 16-byte aligned");
    if (lbh::str60_eq(task, "START")) nbd8_src = nullptr;  // (a new run may reuse the buffer)
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
        PHASE(phase_linesearch(L, flow));
        if (flow == AGAIN) continue;
      }
      PHASE(phase_termination(L, flow));
      PHASE(phase_update(L, flow));
      // (AGAIN and NEXT alike: the next loop trip)
    }
#undef PHASE
  }

  int64_t nfree_g = 0, nenter_g = 0, ileave_g = 0;
  bool index_valid = false;  // a freev has run: wasfree is the membership of Index(1:nfree)

  // ============================================================ state exchange
  int export_state(void *wa_, int32_t *iwa) override {
    // several ranks: every rank exports ITS rows in the same layout (n = n_local); the host matrices
    // are replicated, Index is the local list and -- the global counters of isave not telling how
    // many of THIS rank's rows are free -- Indx2(1) carries the local free count
    if (nranks != 1 && index)
      return fail(LBFGSB_E_STATE, "export_state: contexts that mirror Index are single-rank");
    HIPCHK(hipSetDevice(device));
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
    const T *wa = (const T *)wa_;
    const int64_t mn = (int64_t)m * n;
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
    for (T *dst : {z, r, d, t, xp}) {
      HIPCHK(hipMemcpyAsync(dst, ps, (size_t)n * sizeof(T), hipMemcpyHostToDevice, stream));
      ps += n;
    }
    get(wa8m);
    z_valid = true;  // z as imported
    spec.valid = false, pend.on = 0, pend.impl = 0, d_impl = z_in_x = false, tbrk_valid = false, scan.ready = false;
    nbd8_src = nullptr;
    spcand.valid = false;
    {
      std::vector<lbk::iw_t> h((size_t)n);
      for (int64_t i = 0; i < n; ++i) h[(size_t)i] = (lbk::iw_t)iwa[n + i];
      HIPCHK(hipMemcpy(iwhere, h.data(), (size_t)n * sizeof(lbk::iw_t), hipMemcpyHostToDevice));
    }
    // free-set membership as of the last freev: Index(1:nfree)
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
    if (col < 1 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "bad col/head");
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
        lbk::launch_subsm_update<T>(q, n, 0.5, lean ? (T *)nullptr : z, r_own, pp ? (T *)nullptr : r_own, l, u,
                                    nbd8, iwhere, (const T *)x, (const T *)g, W(), head, col, 1.0, cf, cf,
                                    lean ? (T *)nullptr : d, pp ? (T *)nullptr : t_own, lean ? z : (T *)nullptr, 1,
                                    lbk::Pend{1, 0.5, lean ? 1 : 0}, lean ? t_own : d);
      else             // as the evaluation of a trial point: reduces only
        lbk::launch_update_scan<T>(q, n, (const T *)x, l, u, nbd8, (const T *)g, r_own, lean ? t_own : d,
                                   lean ? 1 : 0, 0.5, iwhere, (T *)nullptr, W(), head, col,
                                   (head + col - 2) % m + 1, 0, 0, nr_flag(col));
    } else
      return fail(LBFGSB_E_ARG, "unknown kernel");
    return 0;
  }
  int k_set_w(const void *hws, const void *hwy) override {
    HIPCHK(hipSetDevice(device));
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
    return 0;
  }
  int k_formk_gram(int col, int head, double *out) override {
    HIPCHK(hipSetDevice(device));
    if (col < 1 || col > m || head < 1 || head > m) return fail(LBFGSB_E_ARG, "formk_gram: bad col/head");
    lbk::launch_formk_gram<T>(q, n, W(), head, col, iwhere);
    const int E = 2 * col * col + col;
    CHK(fetch(E, 0, 0));
    std::memcpy(out, h_res, sizeof(double) * E);
    return 0;
  }
  int k_objective(int kind, const void *x, void *g, double *f) override {
    HIPCHK(hipSetDevice(device));
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
  int sync() override {
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  int attach_rccl(ncclComm_t c, int rank_, int nranks_) override {
    if (comm) g_rccl.CommDestroy(comm);  // (a second init replaces the communicator)
    comm = nullptr;
    CHK(set_ranks(rank_, nranks_));
    comm = c;
    return 0;
  }
  int attach_host(lbfgsb_allreduce_fn ar, lbfgsb_allgather_fn ag, void *user, int rank_,
                  int nranks_) override {
    cb_ar = ar, cb_ag = ag, cb_user = user;
    return set_ranks(rank_, nranks_);
  }
  void path_counts(int64_t &closed_form, int64_t &three_pass) const override {
    closed_form = nclosed, three_pass = nthreepass;
  }
  const void *prev_iterate() const override { return t; }
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
