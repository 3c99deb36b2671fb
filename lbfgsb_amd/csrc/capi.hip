// capi.hip -- the C ABI of include/lbfgsb_hip.h over the type-erased context (solver_base.hpp):
// argument checks, communicator set-up, the minimize loop, the host-pointer form of setulb with
// its handle registry.  No arithmetic here; there is no CPU fallback.
#include "solver_base.hpp"

// ====================================================================== C ABI
extern "C" {

const char *lbfgsb_hip_last_error(void) { return g_err.c_str(); }

int lbfgsb_hip_create(int64_t n_local, int64_t n_global, int64_t row0, int m, int flags,
                      int device, void *stream, lbfgsb_hip_ctx **out) {
  if (!out) return fail(LBFGSB_E_ARG, "out == NULL");
  *out = nullptr;
  if (n_local <= 0 || n_global < n_local || row0 < 0 || row0 + n_local > n_global)
    return fail(LBFGSB_E_ARG, "bad n_local / n_global / row0");
  if (m <= 0 || m > LBFGSB_MAX_M) return fail(LBFGSB_E_ARG, "m must be in 1..LBFGSB_MAX_M");
  // (31 bits: freev's changed-row list keeps a flag in bit 31, Index / Indx2 are exported as int32)
  if (n_local > (int64_t)INT32_MAX - 16) return fail(LBFGSB_E_ARG, "n_local must be < 2^31 - 16 rows per device");
  int rc = 0;
  *out = lbfgsb_make_solver(n_local, n_global, row0, m, flags, device, stream, &rc);
  if (!*out) return rc;
  return LBFGSB_OK;
}

void lbfgsb_hip_destroy(lbfgsb_hip_ctx *ctx) { delete ctx; }

int lbfgsb_hip_rccl_unique_id(void *id128) {
  if (!id128) return fail(LBFGSB_E_ARG, "rccl_unique_id: NULL argument");
  if (!g_rccl.load()) return fail(LBFGSB_E_COMM, "cannot load librccl");
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return fail(LBFGSB_E_COMM, "ncclGetUniqueId");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, 128);
  return 0;
}

int lbfgsb_hip_comm_init_rccl(lbfgsb_hip_ctx *ctx, const void *id128, int rank, int nranks) {
  if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(LBFGSB_E_ARG, "bad rank / NULL id");
  if (!g_rccl.load()) return fail(LBFGSB_E_COMM, "cannot load librccl");
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  // ncclCommInitRank blocks until every rank has arrived; a rank that never comes (a crashed peer, a
  // wrong id) would hang the caller for good.  The call runs on a helper thread and is given
  // LBFGSB_COMM_INIT_TIMEOUT_S seconds (default 120): after that the entry returns LBFGSB_E_COMM -- the
  // helper is left behind, the caller is expected to exit (bench.py does, non-zero)
  double limit = 120.0;
  if (const char *e = std::getenv("LBFGSB_COMM_INIT_TIMEOUT_S")) {
    char *end = nullptr;
    const double v = std::strtod(e, &end);
    if (end == e || *end != '\0' || !(v > 0.0))
      return fail(LBFGSB_E_ARG, std::string("LBFGSB_COMM_INIT_TIMEOUT_S is not a positive number of seconds: '") + e + "'");
    limit = v;
  }
  struct Init {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    bool abandoned = false;  // the caller gave up waiting: a late communicator is destroyed by the helper
    ncclResult_t res = ncclSuccess;
    ncclComm_t comm = nullptr;
  };
  auto st = std::make_shared<Init>();
  const int device = ctx->device;
  std::thread([st, id, rank, nranks, device]() {
    (void)hipSetDevice(device);
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c, nranks, id, rank);
    bool orphan = false;
    {
      std::lock_guard<std::mutex> lk(st->mu);
      st->res = r, st->comm = c, st->done = true;
      orphan = st->abandoned;
      st->cv.notify_all();
    }
    // nobody will ever use (or destroy) a communicator that arrives after the timeout: give it back here, so
    // that it is not leaked and no peer is left with a member that never joins a collective unnoticed
    if (orphan && r == ncclSuccess && c) g_rccl.CommDestroy(c);
  }).detach();
  {
    std::unique_lock<std::mutex> lk(st->mu);
    if (!st->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return st->done; })) {
      st->abandoned = true;
      return fail(LBFGSB_E_COMM, "ncclCommInitRank did not return within " + std::to_string((int)limit) +
                                     " s (a rank is missing?); the process should exit (non-zero): the helper "
                                     "thread is still inside the call");
    }
  }
  if (st->res != ncclSuccess) return fail(LBFGSB_E_COMM, "ncclCommInitRank failed");
  ncclComm_t comm = st->comm;
  // the communicator's own word on its size: what lbfgsb_hip_comm_info reports
  if (g_rccl.CommCount) {
    int cnt = 0;
    if (g_rccl.CommCount(comm, &cnt) != ncclSuccess || cnt != nranks) {
      g_rccl.CommDestroy(comm);
      return fail(LBFGSB_E_COMM, "the communicator reports " + std::to_string(cnt) + " ranks, " +
                                     std::to_string(nranks) + " were asked for");
    }
  }
  const int rc = ctx->attach_rccl(comm, rank, nranks);
  if (rc) g_rccl.CommDestroy(comm);
  return rc;
}

int lbfgsb_hip_comm_info(lbfgsb_hip_ctx *ctx, int32_t *nranks, int32_t *rank, int32_t *kind) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  ncclComm_t c = ctx->rccl_comm();
  int nr = ctx->nranks, rk = ctx->rank;
  if (c && g_rccl.CommCount && g_rccl.CommUserRank) {  // ask the communicator itself
    if (g_rccl.CommCount(c, &nr) != ncclSuccess || g_rccl.CommUserRank(c, &rk) != ncclSuccess)
      return fail(LBFGSB_E_COMM, "ncclCommCount / ncclCommUserRank failed");
  }
  if (nranks) *nranks = nr;
  if (rank) *rank = rk;
  if (kind) *kind = c ? 1 : (ctx->nranks > 1 ? 2 : 0);
  return 0;
}

void *lbfgsb_hip_get_stream(lbfgsb_hip_ctx *ctx) { return ctx ? (void *)ctx->q.stream : nullptr; }

int lbfgsb_hip_wait_stream(lbfgsb_hip_ctx *ctx, void *producer_stream) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  if ((hipStream_t)producer_stream == ctx->q.stream) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->order_ev) HIPCHK(hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(ctx->order_ev, (hipStream_t)producer_stream));
  HIPCHK(hipStreamWaitEvent(ctx->q.stream, ctx->order_ev, 0));
  return 0;
}

int lbfgsb_hip_return_event(lbfgsb_hip_ctx *ctx, void *consumer_stream, int make_wait, void **event_out) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->return_ev) HIPCHK(hipEventCreateWithFlags(&ctx->return_ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(ctx->return_ev, ctx->q.stream));
  if (make_wait && (hipStream_t)consumer_stream != ctx->q.stream)
    HIPCHK(hipStreamWaitEvent((hipStream_t)consumer_stream, ctx->return_ev, 0));
  if (event_out) *event_out = (void *)ctx->return_ev;
  return 0;
}

int lbfgsb_hip_f_device(lbfgsb_hip_ctx *ctx, const void *d_f, void *producer_stream, int order_after) {
  if (!ctx || !d_f) return fail(LBFGSB_E_ARG, "f_device: NULL argument");
  if (order_after) {
    const int rc = lbfgsb_hip_wait_stream(ctx, producer_stream);
    if (rc) return rc;
  }
  return ctx->f_device((const double *)d_f);
}

int lbfgsb_hip_comm_init_host(lbfgsb_hip_ctx *ctx, lbfgsb_allreduce_fn ar, lbfgsb_allgather_fn ag,
                              void *user, int rank, int nranks) {
  if (!ctx || !ar || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(LBFGSB_E_ARG, "bad host reducer arguments");
  return ctx->attach_host(ar, ag, user, rank, nranks);
}

int lbfgsb_hip_setulb_dev(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u,
                          const int32_t *nbd, double *f, void *g, double factr, double pgtol,
                          char *task, int iprint, char *csave, int32_t *lsave, int32_t *isave,
                          double *dsave) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  const int rc = ctx->setulb_dev(x, l, u, nbd, f, g, factr, pgtol, task, iprint, csave, lsave,
                                 isave, dsave);
  if (iprint >= 0) std::fflush(stdout);
  return rc;
}

int lbfgsb_hip_setulb_dev_pp(lbfgsb_hip_ctx *ctx, void *x0, void *x1, const void *l, const void *u,
                             const int32_t *nbd, double *f, void *g0, void *g1, double factr,
                             double pgtol, char *task, int iprint, char *csave, int32_t *lsave,
                             int32_t *isave, double *dsave, int32_t *cur) {
  if (!ctx || !cur) return fail(LBFGSB_E_ARG, "setulb_dev_pp: NULL argument");
  const int rc = ctx->setulb_dev_pp(x0, x1, l, u, nbd, f, g0, g1, factr, pgtol, task, iprint, csave,
                                    lsave, isave, dsave, cur);
  if (iprint >= 0) std::fflush(stdout);
  return rc;
}

int lbfgsb_hip_minimize(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u,
                        const int32_t *nbd, void *g, double factr, double pgtol, int max_iter,
                        int max_fg, int iprint, lbfgsb_fg_fn fg, void *user, int builtin_kind,
                        double *f, char *task, int32_t *lsave, int32_t *isave, double *dsave) {
  if (!ctx || !x || !l || !u || !nbd || !g || !f || !task || !lsave || !isave || !dsave)
    return fail(LBFGSB_E_ARG, "minimize: NULL argument");
  char csave[60];
  std::memset(csave, ' ', 60);
  lbh::str60_set(task, "START");
  for (;;) {
    int rc = ctx->setulb_dev(x, l, u, nbd, f, g, factr, pgtol, task, iprint, csave, lsave, isave,
                             dsave);
    if (rc) return rc;
    if (lbh::str60_pre(task, "FG")) {
      if (fg) {
        rc = ctx->sync();  // a callback may run on any stream
        if (rc) return rc;
        // (the callback must leave g complete or ordered before the context's stream:
        //  lbfgsb_hip_wait_stream / lbfgsb_hip_get_stream, include/lbfgsb_hip.h)
        *f = fg(user, x, g);
        if (std::isnan(*f)) lbh::str60_set(task, "STOP: THE OBJECTIVE CALLBACK RETURNED NaN");
      } else {
        rc = ctx->k_objective(builtin_kind, x, g, nullptr);  // f comes back with the next call
        if (rc) return rc;
      }
    } else if (lbh::str60_pre(task, "NEW_X")) {
      if (max_iter > 0 && isave[29] >= max_iter)
        lbh::str60_set(task, "STOP: MAXIMUM NUMBER OF ITERATIONS REACHED");
      else if (max_fg > 0 && isave[33] >= max_fg)
        lbh::str60_set(task, "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT");
    } else {
      break;
    }
  }
  if (iprint >= 0) std::fflush(stdout);
  return ctx->sync();
}

int lbfgsb_hip_export_state(lbfgsb_hip_ctx *ctx, void *wa, int32_t *iwa) {
  if (!ctx || !wa) return fail(LBFGSB_E_ARG, "export_state: NULL argument");   // (iwa may be NULL: wa only)
  return ctx->export_state(wa, iwa);
}
int lbfgsb_hip_import_state(lbfgsb_hip_ctx *ctx, const void *wa, const int32_t *iwa,
                            const int32_t *isave) {
  if (!ctx || !wa || !iwa || !isave) return fail(LBFGSB_E_ARG, "import_state: NULL argument");
  return ctx->import_state(wa, iwa, isave);
}

int lbfgsb_hip_projgr(lbfgsb_hip_ctx *ctx, const void *x, const void *l, const void *u,
                      const int32_t *nbd, const void *g, double *h_sbgnrm) {
  if (!ctx || !x || !l || !u || !nbd || !g || !h_sbgnrm) return fail(LBFGSB_E_ARG, "projgr: NULL argument");
  return ctx->k_projgr(x, l, u, nbd, g, h_sbgnrm);
}
// ---- routine doors (solver_doors.inl) ----
int lbfgsb_hip_vec_sub(lbfgsb_hip_ctx *ctx, const void *a, const void *b, void *out) {
  if (!ctx || !a || !b || !out) return fail(LBFGSB_E_ARG, "vec_sub: NULL argument");
  return ctx->r_vec_sub(a, b, out);
}
int lbfgsb_hip_vec_scale(lbfgsb_hip_ctx *ctx, double alpha, void *v) {
  if (!ctx || !v) return fail(LBFGSB_E_ARG, "vec_scale: NULL argument");
  return ctx->r_vec_scale(alpha, v);
}
int lbfgsb_hip_dot(lbfgsb_hip_ctx *ctx, const void *a, const void *b, double *h_result) {
  if (!ctx || !a || !b || !h_result) return fail(LBFGSB_E_ARG, "dot: NULL argument");
  return ctx->r_dot(a, b, h_result);
}
int lbfgsb_hip_active(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u, const int32_t *nbd,
                      int32_t *h_flags) {
  if (!ctx || !x || !l || !u || !nbd || !h_flags) return fail(LBFGSB_E_ARG, "active: NULL argument");
  return ctx->r_active(x, l, u, nbd, h_flags);
}
int lbfgsb_hip_errclb(lbfgsb_hip_ctx *ctx, const void *l, const void *u, const int32_t *nbd, double factr,
                      char *task, int32_t *h_info, int64_t *h_k) {
  if (!ctx || !l || !u || !nbd || !task || !h_info || !h_k) return fail(LBFGSB_E_ARG, "errclb: NULL argument");
  return ctx->r_errclb(l, u, nbd, factr, task, h_info, h_k);
}
int lbfgsb_hip_cauchy(lbfgsb_hip_ctx *ctx, const void *x, const void *l, const void *u, const int32_t *nbd,
                      const void *g, double theta, int col, int head, double sbgnrm, void *xcp_out,
                      int32_t *h_nseg, int32_t *h_info) {
  if (!ctx || !x || !l || !u || !nbd || !g || !h_nseg || !h_info)
    return fail(LBFGSB_E_ARG, "cauchy: NULL argument");
  return ctx->r_cauchy(x, l, u, nbd, g, theta, col, head, sbgnrm, xcp_out, h_nseg, h_info);
}
int lbfgsb_hip_freev(lbfgsb_hip_ctx *ctx, int iter, int cnstnd, int updatd, int64_t *h_nfree,
                     int64_t *h_nenter, int64_t *h_ileave, int32_t *h_wrk) {
  if (!ctx || !h_nfree || !h_nenter || !h_ileave || !h_wrk) return fail(LBFGSB_E_ARG, "freev: NULL argument");
  return ctx->r_freev(iter, cnstnd, updatd, h_nfree, h_nenter, h_ileave, h_wrk);
}
int lbfgsb_hip_formk(lbfgsb_hip_ctx *ctx, int col, int head, double theta, int32_t *h_info) {
  if (!ctx || !h_info) return fail(LBFGSB_E_ARG, "formk: NULL argument");
  return ctx->r_formk(col, head, theta, h_info);
}
int lbfgsb_hip_cmprlb(lbfgsb_hip_ctx *ctx, const void *x, const void *g, double theta, int col, int head,
                      int cnstnd, void *r_out, int32_t *h_info) {
  if (!ctx || !x || !g || !h_info) return fail(LBFGSB_E_ARG, "cmprlb: NULL argument");
  return ctx->r_cmprlb(x, g, theta, col, head, cnstnd, r_out, h_info);
}
int lbfgsb_hip_subsm(lbfgsb_hip_ctx *ctx, const void *x, const void *l, const void *u, const int32_t *nbd,
                     const void *g, const void *r_in, double theta, int col, int head, void *xhat_out,
                     int32_t *h_iword, int32_t *h_info) {
  if (!ctx || !x || !l || !u || !nbd || !g || !r_in || !h_iword || !h_info)
    return fail(LBFGSB_E_ARG, "subsm: NULL argument");
  return ctx->r_subsm(x, l, u, nbd, g, r_in, theta, col, head, xhat_out, h_iword, h_info);
}
int lbfgsb_hip_lnsrlb(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u, const int32_t *nbd,
                      const void *g, double f, double *h_sc, int32_t *h_ic, char *task, char *csave,
                      int32_t *h_isave2, double *h_dsave13) {
  if (!ctx || !x || !l || !u || !nbd || !g || !h_sc || !h_ic || !task || !csave || !h_isave2 || !h_dsave13)
    return fail(LBFGSB_E_ARG, "lnsrlb: NULL argument");
  return ctx->r_lnsrlb(x, l, u, nbd, g, f, h_sc, h_ic, task, csave, h_isave2, h_dsave13);
}
int lbfgsb_hip_matupd(lbfgsb_hip_ctx *ctx, const void *g, double stp, double dr, double dtd, int32_t *h_ip,
                      double *h_theta) {
  if (!ctx || !g || !h_ip || !h_theta) return fail(LBFGSB_E_ARG, "matupd: NULL argument");
  return ctx->r_matupd(g, stp, dr, dtd, h_ip, h_theta);
}

int lbfgsb_hip_wtv(lbfgsb_hip_ctx *ctx, const void *v, int col, int head, double *h_out) {
  if (!ctx || !v || !h_out) return fail(LBFGSB_E_ARG, "wtv: NULL argument");
  return ctx->k_wtv(v, col, head, h_out, false);
}
int lbfgsb_hip_wtv_launch_only(lbfgsb_hip_ctx *ctx, const void *v, int col, int head) {
  if (!ctx || !v) return fail(LBFGSB_E_ARG, "wtv: NULL argument");
  return ctx->k_wtv(v, col, head, nullptr, true);
}
int lbfgsb_hip_wtv_time(lbfgsb_hip_ctx *ctx, const void *v, int col, int head, int reps,
                        double *h_ms_per_launch) {
  if (!ctx || !v || !h_ms_per_launch || reps < 1) return fail(LBFGSB_E_ARG, "wtv_time: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k) CHK(ctx->k_wtv(v, col, head, nullptr, true));
  HIPCHK(hipEventRecord(e0, ctx->q.stream));
  for (int k = 0; k < reps; ++k) CHK(ctx->k_wtv(v, col, head, nullptr, true));
  HIPCHK(hipEventRecord(e1, ctx->q.stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *h_ms_per_launch = (double)ms / reps;
  return 0;
}
int lbfgsb_hip_kernel_time(lbfgsb_hip_ctx *ctx, int which, const void *x, const void *g, int col,
                           int head, int reps, double *h_ms_per_launch) {
  if (!ctx || !x || !g || !h_ms_per_launch || reps < 1) return fail(LBFGSB_E_ARG, "kernel_time: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  for (int k = 0; k < 2; ++k) CHK(ctx->k_launch(which, x, g, col, head));
  HIPCHK(hipEventRecord(e0, ctx->q.stream));
  for (int k = 0; k < reps; ++k) CHK(ctx->k_launch(which, x, g, col, head));
  HIPCHK(hipEventRecord(e1, ctx->q.stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *h_ms_per_launch = (double)ms / reps;
  return 0;
}
int lbfgsb_hip_set_w(lbfgsb_hip_ctx *ctx, const void *h_ws, const void *h_wy) {
  if (!ctx || !h_ws || !h_wy) return fail(LBFGSB_E_ARG, "set_w: NULL argument");
  return ctx->k_set_w(h_ws, h_wy);
}
int lbfgsb_hip_set_iwhere(lbfgsb_hip_ctx *ctx, const int32_t *h_iwhere) {
  if (!ctx || !h_iwhere) return fail(LBFGSB_E_ARG, "set_iwhere: NULL argument");
  return ctx->k_set_iwhere(h_iwhere);
}
int lbfgsb_hip_formk_gram(lbfgsb_hip_ctx *ctx, int col, int head, double *h_out) {
  if (!ctx || !h_out) return fail(LBFGSB_E_ARG, "formk_gram: NULL argument");
  return ctx->k_formk_gram(col, head, h_out);
}
int lbfgsb_hip_sync(lbfgsb_hip_ctx *ctx) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  return ctx->sync();
}
int lbfgsb_hip_objective(lbfgsb_hip_ctx *ctx, int kind, const void *x, void *g, double *h_f) {
  if (!ctx || !x || !g) return fail(LBFGSB_E_ARG, "objective: NULL argument");   // (h_f may be NULL: deferred)
  return ctx->k_objective(kind, x, g, h_f);
}
int lbfgsb_hip_stats(lbfgsb_hip_ctx *ctx, int64_t *launches, int64_t *syncs,
                     int64_t *cauchy_fullsorts, double *wait_seconds) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  if (launches) *launches = ctx->q.launches;
  if (syncs) *syncs = ctx->nsync;
  if (cauchy_fullsorts) *cauchy_fullsorts = ctx->nfullsort;
  if (wait_seconds) *wait_seconds = ctx->t_wait;
  return 0;
}

int lbfgsb_hip_comm_stats(lbfgsb_hip_ctx *ctx, int64_t *collectives, int64_t *bytes_contributed) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  if (collectives) *collectives = ctx->ncoll;
  if (bytes_contributed) *bytes_contributed = ctx->coll_bytes;
  return 0;
}

int lbfgsb_hip_freev_skipped(lbfgsb_hip_ctx *ctx, int64_t *count) {
  if (!ctx || !count) return fail(LBFGSB_E_ARG, "freev_skipped: NULL argument");
  *count = ctx->freev_skipped();
  return 0;
}

int lbfgsb_hip_skip_stats(lbfgsb_hip_ctx *ctx, int64_t *scans_reused) {
  if (!ctx || !scans_reused) return fail(LBFGSB_E_ARG, "skip_stats: NULL argument");
  *scans_reused = ctx->skip_scans_reused();
  return 0;
}

int lbfgsb_hip_uniform_bounds(lbfgsb_hip_ctx *ctx, int32_t *mask) {
  if (!ctx || !mask) return fail(LBFGSB_E_ARG, "uniform_bounds: NULL argument");
  *mask = ctx->uniform_mask();
  return 0;
}

int lbfgsb_hip_path_counts(lbfgsb_hip_ctx *ctx, int64_t *closed_form, int64_t *three_pass,
                           int64_t *handed_windows) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  if (handed_windows) *handed_windows = ctx->nspecwin;
  int64_t a = 0, b = 0;
  ctx->path_counts(a, b);
  if (closed_form) *closed_form = a;
  if (three_pass) *three_pass = b;
  return 0;
}

int lbfgsb_hip_set_option(lbfgsb_hip_ctx *ctx, const char *name, double value) {
  if (!ctx || !name) return fail(LBFGSB_E_ARG, "set_option: NULL argument");
  return ctx->set_option(name, value);
}

int lbfgsb_hip_tie_splits(lbfgsb_hip_ctx *ctx, int64_t *count) {
  if (!ctx || !count) return fail(LBFGSB_E_ARG, "tie_splits: NULL argument");
  *count = ctx->ntiesplit;
  return 0;
}

int lbfgsb_hip_collective_time(lbfgsb_hip_ctx *ctx, int reps, double *median_us, double *min_us) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  return ctx->collective_time(reps, median_us, min_us);
}

int lbfgsb_hip_compact_stats(lbfgsb_hip_ctx *ctx, int64_t *packs, int64_t *unpacks, int32_t *packed,
                              int32_t *eligible) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  int64_t a = 0, b = 0;
  int c = 0, d = 0;
  ctx->compact_stats(a, b, c, d);
  if (packs) *packs = a;
  if (unpacks) *unpacks = b;
  if (packed) *packed = c;
  if (eligible) *eligible = d;
  return 0;
}

int lbfgsb_hip_refresh_count(lbfgsb_hip_ctx *ctx, int64_t *count) {
  if (!ctx || !count) return fail(LBFGSB_E_ARG, "refresh_count: NULL argument");
  *count = ctx->nrefresh;
  return 0;
}

int lbfgsb_hip_host_gap(lbfgsb_hip_ctx *ctx, double *seconds, int64_t *count) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  if (seconds) *seconds = ctx->t_mid;
  if (count) *count = ctx->n_mid;
  return 0;
}
int lbfgsb_hip_host_segments(lbfgsb_hip_ctx *ctx, double *seconds5) {
  if (!ctx || !seconds5) return fail(LBFGSB_E_ARG, "host_segments: NULL argument");
  for (int k = 0; k < 5; ++k) seconds5[k] = ctx->t_seg[k];
  return 0;
}

int lbfgsb_hip_defer_stats(lbfgsb_hip_ctx *ctx, int64_t *deferred, int64_t *reissued) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  int64_t a = 0, b = 0;
  ctx->defer_counts(a, b);
  if (deferred) *deferred = a;
  if (reissued) *reissued = b;
  return 0;
}

int lbfgsb_hip_pass_clock(lbfgsb_hip_ctx *ctx, int enable, double *ms_total, int64_t *count) {
  if (!ctx) return fail(LBFGSB_E_ARG, "ctx == NULL");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->clk_stream = ctx->q.stream;
  if (enable == 1) {
    for (int k = 0; k < 3; ++k) {
      ctx->clk_ms[k] = 0.0, ctx->clk_n[k] = 0;
      for (bool &pnd : ctx->clk_pending[k]) pnd = false;
    }
    ctx->clk_dropped = 0;
    ctx->clock_on = true;
  } else {
    HIPCHK(hipStreamSynchronize(ctx->q.stream));
    ctx->clk_collect();
    if (enable == 0) ctx->clock_on = false;
  }
  for (int k = 0; k < 3; ++k) {
    if (ms_total) ms_total[k] = ctx->clk_ms[k];
    if (count) count[k] = ctx->clk_n[k];
  }
  return 0;
}

// ----------------------------------------------------------- host-pointer form
// The exact reference signature (src/lbfgsb.f90:88-89) plus real_bytes/mirror.  The context of a
// run is kept in a process-wide registry; isave(17:18) hold {id, tag} -- slots the reference
// never writes (:250-284) -- never a raw pointer.
namespace {
constexpr int32_t HOST_TAG = 0x4C424642;  // "LBFB"
struct HostRegistry {
  std::mutex mu;
  std::unordered_map<int32_t, lbfgsb_hip_ctx *> live;
  int32_t next_id = 1;
  // Contexts of runs that were abandoned without lbfgsb_hip_release_host (the reference's driver2 /
  // driver3 leave with task = 'STOP' and never re-enter setulb) are NOT destroyed from this static
  // object's destructor: hipFree / hipStreamDestroy / ncclCommDestroy at process exit are unordered
  // against the HIP runtime's own teardown and can fault or hang.  The process exit reclaims them.
  ~HostRegistry() { live.clear(); }
  int32_t add(lbfgsb_hip_ctx *c) {
    std::lock_guard<std::mutex> lk(mu);
    while (live.count(next_id) || next_id <= 0) next_id = next_id == INT32_MAX ? 1 : next_id + 1;
    live[next_id] = c;
    return next_id;
  }
  lbfgsb_hip_ctx *find(const int32_t *isave) {
    if (isave[17] != HOST_TAG) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(isave[16]);
    return it == live.end() ? nullptr : it->second;
  }
  void drop(int32_t *isave) {
    lbfgsb_hip_ctx *c = nullptr;
    if (isave[17] == HOST_TAG) {
      std::lock_guard<std::mutex> lk(mu);
      auto it = live.find(isave[16]);
      if (it != live.end()) {
        c = it->second;
        live.erase(it);
      }
    }
    delete c;
    isave[16] = isave[17] = 0;
  }
};
HostRegistry g_host;
std::atomic<int> g_host_pinning{1};  // lbfgsb_hip_host_pinning

// Runs the caller abandoned (the reference's driver2 / driver3 set task = 'STOP' and never re-enter setulb) keep
// their context and the registrations of the arrays they pinned.  A START whose arrays overlap such a registration
// can only mean that run is over (the same arrays started again, or the memory was freed and handed out again):
// it is dropped here, its registrations with it, before the new run pins anything.
void drop_runs_overlapping(const void *const *ptr, const size_t *bytes, int cnt) {
  std::vector<lbfgsb_hip_ctx *> dead;
  {
    std::lock_guard<std::mutex> lk(g_host.mu);
    for (auto it = g_host.live.begin(); it != g_host.live.end();) {
      bool hit = false;
      for (const auto &r : it->second->host_reg)
        for (int k = 0; k < cnt && !hit; ++k)
          hit = r.p && ptr[k] && (const char *)ptr[k] < (const char *)r.p + r.bytes &&
                (const char *)r.p < (const char *)ptr[k] + bytes[k];
      if (hit) {
        dead.push_back(it->second);
        it = g_host.live.erase(it);
      } else {
        ++it;
      }
    }
  }
  for (lbfgsb_hip_ctx *c : dead) delete c;
}
}  // namespace

int lbfgsb_hip_host_pinning(int mode) {
  if (mode != 0 && mode != 1) return fail(LBFGSB_E_ARG, "host_pinning: mode must be 0 or 1");
  g_host_pinning.store(mode);
  return 0;
}

int lbfgsb_hip_release_host(int32_t *isave) {
  if (!isave) return fail(LBFGSB_E_ARG, "isave == NULL");
  g_host.drop(isave);
  return 0;
}

// One implementation for both integer widths of the caller (int_bytes = 4: default integers of an
// ordinary Fortran build; 8: -fdefault-integer-8, the build BASELINE.md section 3 calls mandatory for
// n = 1e8 because the reference's own wa offsets overflow int32 there, src/lbfgsb.f90:246-265).  nbd,
// iwa, lsave, isave are arrays of that width; inside, the library works with int32 copies (nbd is
// narrowed once, on START; isave / lsave are 44 + 4 words per call; iwa only travels when mirror != 0).
static int setulb_host_impl(int64_t n, int64_t m, void *x, const void *l, const void *u, const void *nbd_,
                            void *f, void *g, double factr, double pgtol, void *wa, void *iwa_, char *task,
                            int32_t iprint, char *csave, void *lsave_, void *isave_, void *dsave,
                            const char *iteration_file, int32_t real_bytes, int32_t mirror, int int_bytes) {
  const bool r32 = real_bytes == 4;
  if (!x || !l || !u || !nbd_ || !f || !g || !task || !csave || !lsave_ || !isave_ || !dsave)
    return fail(LBFGSB_E_ARG, "setulb: NULL argument");
  if (mirror && (!wa || !iwa_)) return fail(LBFGSB_E_ARG, "setulb: mirror = 1 needs wa and iwa");
  if (real_bytes != 4 && real_bytes != 8) return fail(LBFGSB_E_ARG, "real_bytes must be 4 or 8");
  if (int_bytes != 4 && int_bytes != 8) return fail(LBFGSB_E_ARG, "int_bytes must be 4 or 8");
  const bool i8 = int_bytes == 8;
  const size_t rb = (size_t)real_bytes;
  // int32 views of the caller's small integer arrays
  int32_t isave[44], lsave[4];
  auto geti = [&](const void *p, int k) -> int64_t {
    return i8 ? ((const int64_t *)p)[k] : (int64_t)((const int32_t *)p)[k];
  };
  auto puti = [&](void *p, int k, int64_t v) {
    if (i8)
      ((int64_t *)p)[k] = v;
    else
      ((int32_t *)p)[k] = (int32_t)std::min<int64_t>(std::max<int64_t>(v, INT32_MIN), INT32_MAX);
  };
  for (int k = 0; k < 44; ++k) {
    const int64_t v = geti(isave_, k);
    isave[k] = (int32_t)std::min<int64_t>(std::max<int64_t>(v, INT32_MIN), INT32_MAX);
  }
  for (int k = 0; k < 4; ++k) lsave[k] = geti(lsave_, k) != 0;
  int64_t off64[16] = {0};
  bool have_off = false;
  auto finish_ints = [&]() {
    for (int k = 0; k < 44; ++k) puti(isave_, k, isave[k]);
    // the wa offsets of isave(1:16) in full width for a 64-bit caller (saturated for a 32-bit one)
    if (have_off)
      for (int k = 0; k < 16; ++k) puti(isave_, k, off64[k]);
    for (int k = 0; k < 4; ++k) puti(lsave_, k, lsave[k]);
  };
  lbfgsb_hip_ctx *ctx = nullptr;
  const bool start = lbh::str60_eq(task, "START");
  if (start) {
    g_host.drop(isave);  // a START over the isave of a run that is still registered
    // the reference's own argument checks that do not need a context (:1618-1620)
    if (n <= 0 || m <= 0) {
      if (n <= 0) lbh::str60_set(task, "ERROR: N <= 0");
      if (m <= 0) lbh::str60_set(task, "ERROR: M <= 0");
      finish_ints();
      return 0;
    }
    // The reference puts no upper limit on m (:93-97); this library takes up to LBFGSB_MAX_M pairs (fused
    // kernels up to LBFGSB_FUSED_M, unfused tiles beyond).  A caller written against the reference sees the
    // limit the way it sees every other argument error: a task that starts with 'ERROR', no iteration done.
    if (m > LBFGSB_MAX_M) {
      lbh::str60_set(task, "ERROR: M > 1024 (LIMIT OF LBFGSB_HIP)");
      finish_ints();
      return 0;
    }
    // (row numbers of one device travel in 31 bits: freev's changed-row list keeps a flag in bit 31,
    //  Index / Indx2 are exported as int32)
    if (n > (int64_t)INT32_MAX - 16) {
      lbh::str60_set(task, "ERROR: N >= 2**31 ON ONE DEVICE (LIMIT OF LBFGSB_HIP)");
      finish_ints();
      return 0;
    }
    // (NO_RETURN_SYNC: this function queues its own D2H copies behind the call and synchronises once)
    int fl = (r32 ? LBFGSB_F_REAL32 : 0) | (mirror ? LBFGSB_F_MIRROR_INDEX : 0) | LBFGSB_F_NO_RETURN_SYNC;
    int rc = lbfgsb_hip_create(n, n, 0, (int)m, fl, 0, nullptr, &ctx);
    if (rc) return rc;
    // the caller's x, g and the t slot of wa are pinned for the run: the per-call transfers below are then DMA
    // copies on the context's stream (unpinned again when the context goes; arrays that cannot be pinned, or
    // other arrays than these on a later call, travel as pageable copies)
    // lbfgsb_hip_host_pinning(0) switches this off for the process (include/lbfgsb_hip.h: when to).
    if (g_host_pinning.load()) {
      void *tslot = (wa && !mirror) ? (char *)wa + (size_t)(2ll * m * n + 11ll * m * m + 3ll * n) * rb : nullptr;
      const void *pv[3] = {x, g, tslot};
      const size_t pb[3] = {(size_t)n * rb, (size_t)n * rb, (size_t)n * rb};
      drop_runs_overlapping(pv, pb, 3);
      ctx->host_register(0, x, pb[0]);
      ctx->host_register(1, g, pb[1]);
      if (tslot) ctx->host_register(2, tslot, pb[2]);
    }
    if (iteration_file && iteration_file[0]) ctx->itfile_name = iteration_file;
    const size_t vb = ((size_t)n + 32) * rb;
    auto stage = [&]() -> int {
      HIPCHK(hipMalloc(&ctx->hx, vb));
      HIPCHK(hipMalloc(&ctx->hg, vb));
      HIPCHK(hipMalloc(&ctx->hl, vb));
      HIPCHK(hipMalloc(&ctx->hu, vb));
      HIPCHK(hipMalloc(&ctx->hnbd, ((size_t)n + 32) * 4));
      HIPCHK(hipMemset(ctx->hg, 0, vb));
      HIPCHK(hipMemcpy(ctx->hx, x, (size_t)n * rb, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(ctx->hl, l, (size_t)n * rb, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(ctx->hu, u, (size_t)n * rb, hipMemcpyHostToDevice));
      if (i8) {  // (values outside int32 are invalid bound types anyway: errclb reports them)
        std::vector<int32_t> nb((size_t)n);
        const int64_t *src = (const int64_t *)nbd_;
        for (int64_t k = 0; k < n; ++k)
          nb[(size_t)k] = (src[k] < 0 || src[k] > 3) ? -1 : (int32_t)src[k];
        HIPCHK(hipMemcpy(ctx->hnbd, nb.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      } else {
        HIPCHK(hipMemcpy(ctx->hnbd, nbd_, (size_t)n * 4, hipMemcpyHostToDevice));
      }
      return 0;
    };
    rc = stage();
    if (rc) {
      lbfgsb_hip_destroy(ctx);
      return rc;
    }
    std::memset(isave, 0, sizeof isave);
    isave[16] = g_host.add(ctx);
    isave[17] = HOST_TAG;
  } else {
    ctx = g_host.find(isave);
    if (!ctx || ctx->n != n || ctx->m != m)
      return fail(LBFGSB_E_STATE, "setulb called without a live context (task must be START first)");
    // a caller that hands over OTHER arrays than START's (a fresh g per call, say): the START-time array may be
    // gone already -- its registration is released now, this call's array travels as a pageable copy
    ctx->host_unregister_if_not(0, x);
    ctx->host_unregister_if_not(1, g);
    if (wa && !mirror)
      ctx->host_unregister_if_not(2, (char *)wa + (size_t)(2ll * m * n + 11ll * m * m + 3ll * n) * rb);
    // l, u, nbd were copied at START; the reference re-reads the caller's arrays on every call (:1270-1330,
    // :2594-2622, :2789-2816).  After the first iteration and then every 32nd the arrays are uploaded again and
    // compared with the copies, bit for bit: an edit in place ends the run with an error instead of being ignored.
    if (lbh::str60_pre(task, "NEW_X") && (isave[29] == 1 || (isave[29] > 0 && isave[29] % 32 == 0))) {
      void *tl = nullptr, *tu = nullptr;
      int32_t *tn = nullptr;
      double ndiff = 0.0;
      auto check = [&]() -> int {
        const size_t vb = ((size_t)n + 32) * rb;
        HIPCHK(hipMalloc(&tl, vb));
        HIPCHK(hipMalloc(&tu, vb));
        HIPCHK(hipMalloc(&tn, ((size_t)n + 32) * 4));
        HIPCHK(hipMemcpy(tl, l, (size_t)n * rb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(tu, u, (size_t)n * rb, hipMemcpyHostToDevice));
        if (i8) {
          std::vector<int32_t> nb((size_t)n);
          const int64_t *src = (const int64_t *)nbd_;
          for (int64_t k = 0; k < n; ++k) nb[(size_t)k] = (src[k] < 0 || src[k] > 3) ? -1 : (int32_t)src[k];
          HIPCHK(hipMemcpy(tn, nb.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        } else {
          HIPCHK(hipMemcpy(tn, nbd_, (size_t)n * 4, hipMemcpyHostToDevice));
        }
        return ctx->bounds_same(ctx->hl, ctx->hu, ctx->hnbd, tl, tu, tn, &ndiff);
      };
      int rcc = check();
      if (tl) (void)hipFree(tl);
      if (tu) (void)hipFree(tu);
      if (tn) (void)hipFree(tn);
      if (rcc && (!tl || !tu || !tn)) {
        // no room for the three temporary copies (a run that fits must not fail for a CHECK): skipped this time
        (void)hipGetLastError();
        rcc = 0, ndiff = 0.0;
      }
      if (rcc) return rcc;
      if (ndiff != 0.0) {
        lbh::str60_set(task, "ERROR: BOUNDS CHANGED DURING RUN");
        if (iprint >= 0) std::printf("\n l, u or nbd were modified during the run (%.0f rows differ from the arrays of task = 'START').\n", ndiff);
        g_host.drop(isave);
        finish_ints();
        return 0;
      }
    }
    // the caller's gradient (and f) at the point the last return asked for; x is the library's own trial point
    // and is not read back (the reference's caller does not modify x between calls either, :104-108)
    if (lbh::str60_pre(task, "FG"))
      HIPCHK(hipMemcpyAsync(ctx->hg, g, (size_t)n * rb, hipMemcpyHostToDevice, ctx->q.stream));
  }
  // the wa offsets the reference persists in isave(4:16) (:250-265: lws, lwy, lsy, lss, lwt, lwn,
  // lsnd, lz, lr, ld, lt, lxp, lwa; isave(1:3) = m*n, m^2, 4m^2), 1-based, computed in 64 bits
  // (the reference's default-integer arithmetic wraps at n = 1e8, m = 10 in a 32-bit build)
  {
    const int64_t mn = m * n, mm = m * m;
    const int64_t lws = 1, lwy = lws + mn, lsy = lwy + mn, lss = lsy + mm, lwt = lss + mm,
                  lwn = lwt + mm, lsnd = lwn + 4 * mm, lz = lsnd + 4 * mm, lr = lz + n, ld_ = lr + n,
                  lt = ld_ + n, lxp = lt + n, lwa = lxp + n;
    const int64_t v[16] = {mn, mm, 4 * mm, lws, lwy, lsy, lss, lwt, lwn, lsnd, lz, lr, ld_, lt, lxp, lwa};
    std::memcpy(off64, v, sizeof off64);
    have_off = true;
  }
  const int32_t keep_id = isave[16], keep_tag = isave[17];
  double fd = r32 ? (double)*(float *)f : *(double *)f;
  double ds[29];
  for (int i = 0; i < 29; ++i) ds[i] = r32 ? (double)((float *)dsave)[i] : ((double *)dsave)[i];
  ctx->restored_xg = false;
  const int64_t setups0 = ctx->n_ls_setup;
  int rc = ctx->setulb_dev(ctx->hx, ctx->hl, ctx->hu, ctx->hnbd, &fd, ctx->hg, factr, pgtol, task,
                           iprint, csave, lsave, isave, ds);
  isave[16] = keep_id, isave[17] = keep_tag;
  if (iprint >= 0) std::fflush(stdout);
  if (rc) {
    // a START that failed after its context was registered: the caller's isave never received the
    // handle, so nothing could release the context later -- free it here
    if (start) g_host.drop(isave);
    return rc;
  }
  for (int i = 0; i < 29; ++i) {
    if (r32)
      ((float *)dsave)[i] = (float)ds[i];
    else
      ((double *)dsave)[i] = ds[i];
  }
  if (r32)
    *(float *)f = (float)fd;
  else
    *(double *)f = fd;
  // What travels back (the reference writes these arrays only at these points):
  //   x   at START (active's projection, :965-1040), at every 'FG_LNSRCH' return (the trial point, :2262-2268)
  //       and when the call restored the previous iterate (:568-569, :736-737) -- at 'NEW_X' and at the
  //       convergence returns x is the point the caller already holds;
  //   g   only when the call restored it (the same two places); otherwise the device copy IS the caller's;
  //   t   (wa's previous-iterate slot) only when a line search was set up in this call (:2235).
  // Queued on the context's stream behind the call's kernels, one synchronisation for all of them.
  hipStream_t st = ctx->q.stream;
  const bool x_back = start || ctx->restored_xg || lbh::str60_pre(task, "FG_LN");
  if (x_back) HIPCHK(hipMemcpyAsync(x, ctx->hx, (size_t)n * rb, hipMemcpyDeviceToHost, st));
  if (ctx->restored_xg) HIPCHK(hipMemcpyAsync(g, ctx->hg, (size_t)n * rb, hipMemcpyDeviceToHost, st));
  if (mirror) {
    HIPCHK(hipStreamSynchronize(st));
    if (i8) {  // the library's int32 iwa, widened into the caller's
      std::vector<int32_t> iw((size_t)3 * (size_t)n);
      rc = ctx->export_state(wa, iw.data());
      if (rc) return rc;
      int64_t *dst = (int64_t *)iwa_;
      for (size_t k = 0; k < iw.size(); ++k) dst[k] = iw[k];
    } else {
      rc = ctx->export_state(wa, (int32_t *)iwa_);
      if (rc) return rc;
    }
  } else if (wa && ctx->n_ls_setup != setups0) {
    // previous iterate: wa(3n+2mn+11m^2+1 : +n), read by test/driver3.f90:171-175
    const int64_t off_t = 2ll * m * n + 11ll * m * m + 3ll * n;
    const void *src = ctx->prev_iterate();
    HIPCHK(hipMemcpyAsync((char *)wa + (size_t)off_t * rb, src, (size_t)n * rb, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  if (!lbh::str60_pre(task, "FG") && !lbh::str60_pre(task, "NEW_X"))
    g_host.drop(isave);  // terminal task: CONVERGENCE / ABNORMAL / ERROR / STOP
  finish_ints();
  return 0;
}

int lbfgsb_hip_setulb_host(int32_t n, int32_t m, void *x, const void *l, const void *u,
                           const int32_t *nbd, void *f, void *g, double factr, double pgtol,
                           void *wa, int32_t *iwa, char *task, int32_t iprint, char *csave,
                           int32_t *lsave, int32_t *isave, void *dsave, const char *iteration_file,
                           int32_t real_bytes, int32_t mirror) {
  return setulb_host_impl(n, m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, task, iprint, csave, lsave, isave,
                          dsave, iteration_file, real_bytes, mirror, 4);
}

int lbfgsb_hip_setulb_host_ik(int64_t n, int64_t m, void *x, const void *l, const void *u, const void *nbd,
                              void *f, void *g, double factr, double pgtol, void *wa, void *iwa,
                              char *task, int64_t iprint, char *csave, void *lsave, void *isave,
                              void *dsave, const char *iteration_file, int32_t real_bytes,
                              int32_t mirror, int32_t int_bytes) {
  const int32_t ipr = (int32_t)std::min<int64_t>(std::max<int64_t>(iprint, -1), 1000);
  return setulb_host_impl(n, m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, task, ipr, csave, lsave, isave,
                          dsave, iteration_file, real_bytes, mirror, int_bytes);
}

int lbfgsb_hip_release_host_ik(void *isave, int32_t int_bytes) {
  if (!isave || (int_bytes != 4 && int_bytes != 8)) return fail(LBFGSB_E_ARG, "release_host: bad argument");
  int32_t tmp[44] = {0};
  for (int k = 16; k < 18; ++k)
    tmp[k] = int_bytes == 8 ? (int32_t)((int64_t *)isave)[k] : ((int32_t *)isave)[k];
  g_host.drop(tmp);
  for (int k = 16; k < 18; ++k) {
    if (int_bytes == 8)
      ((int64_t *)isave)[k] = 0;
    else
      ((int32_t *)isave)[k] = 0;
  }
  return 0;
}

}  // extern "C"
