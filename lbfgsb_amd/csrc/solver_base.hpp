// solver_base.hpp -- what solver.hip (the mainlb state machine, Solver<T>) and capi.hip (the C ABI
// of include/lbfgsb_hip.h) share: error plumbing, the dlopen'ed RCCL entry points, and the
// type-erased context the C ABI hands out.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lbfgsb_hip.h"
#include "../../include/lbfgsb_hip_debug.h"
#include "host_dense.hpp"
#include "kernels.hpp"
#include "report.hpp"

namespace lbs {

inline thread_local std::string g_err;

inline int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(LBFGSB_E_NOGPU, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)
#define CHK(expr)          \
  do {                     \
    int rc_ = (expr);      \
    if (rc_ != 0) return rc_; \
  } while (0)

inline double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- RCCL (dlopen)
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  // (optional: what a communicator says about itself, lbfgsb_hip_comm_info)
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  bool ok = false;  // every required symbol resolved
  std::mutex mtx;   // (contexts are created from any number of host threads)
  bool load() {
    std::lock_guard<std::mutex> lock(mtx);
    if (ok) return true;
    if (h) {  // an earlier attempt found a library without the symbols: try again from scratch
      dlclose(h);
      h = nullptr;
    }
    // LBFGSB_RCCL_LIBRARY: a specific build of the library (tests point it at a small
    // shared-memory stand-in so that the communicator code path runs with several ranks on one GPU)
    const char *names[] = {std::getenv("LBFGSB_RCCL_LIBRARY"), "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
      if (!nm || !*nm) continue;
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return false;
#define SYM(f, name) f = reinterpret_cast<decltype(f)>(dlsym(h, name))
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
#undef SYM
    ok = GetUniqueId && CommInitRank && CommDestroy && AllGather;
    if (!ok) {
      dlclose(h);
      h = nullptr;
    }
    return ok;
  }
};
inline Rccl g_rccl;

struct Rec {  // one breakpoint as the host walk needs it
  double t;
  int64_t gidx;
};

}  // namespace lbs
using namespace lbs;

// ===================================================================== context
struct lbfgsb_hip_ctx {
  virtual ~lbfgsb_hip_ctx() { host_unregister_all(); }
  virtual int setulb_dev(void *x, const void *l, const void *u, const int32_t *nbd, double *f,
                         void *g, double factr, double pgtol, char *task, int iprint, char *csave,
                         int32_t *lsave, int32_t *isave, double *dsave) = 0;
  virtual int setulb_dev_pp(void *x0, void *x1, const void *l, const void *u, const int32_t *nbd,
                            double *f, void *g0, void *g1, double factr, double pgtol, char *task,
                            int iprint, char *csave, int32_t *lsave, int32_t *isave, double *dsave,
                            int32_t *cur) = 0;
  virtual int export_state(void *wa, int32_t *iwa) = 0;
  virtual int import_state(const void *wa, const int32_t *iwa, const int32_t *isave) = 0;
  virtual int k_projgr(const void *x, const void *l, const void *u, const int32_t *nbd,
                       const void *g, double *out) = 0;
  virtual int k_wtv(const void *v, int col, int head, double *out, bool launch_only) = 0;
  virtual int k_set_w(const void *hws, const void *hwy) = 0;
  virtual int k_set_iwhere(const int32_t *h_iw) = 0;
  virtual int k_formk_gram(int col, int head, double *out) = 0;
  virtual int k_launch(int which, const void *x, const void *g, int col, int head) = 0;
  virtual int k_objective(int kind, const void *x, void *g, double *f) = 0;
  // routine doors (solver_doors.inl): one routine of the reference each, on the state of the context
  virtual int r_active(void *x, const void *l, const void *u, const int32_t *nbd, int32_t *out3) = 0;
  virtual int r_vec_sub(const void *a, const void *b, void *out) = 0;
  virtual int r_vec_scale(double alpha, void *v) = 0;
  virtual int r_dot(const void *a, const void *b, double *out) = 0;
  virtual int r_errclb(const void *l, const void *u, const int32_t *nbd, double factr, char *task,
                       int32_t *info, int64_t *k) = 0;
  virtual int r_cauchy(const void *x, const void *l, const void *u, const int32_t *nbd, const void *g,
                       double theta, int col, int head, double sbgnrm, void *xcp_out, int32_t *nseg,
                       int32_t *info) = 0;
  virtual int r_freev(int iter, int cnstnd, int updatd, int64_t *nfree, int64_t *nenter, int64_t *ileave,
                      int32_t *wrk) = 0;
  virtual int r_formk(int col, int head, double theta, int32_t *info) = 0;
  virtual int r_cmprlb(const void *x, const void *g, double theta, int col, int head, int cnstnd, void *r_out,
                       int32_t *info) = 0;
  virtual int r_subsm(const void *x, const void *l, const void *u, const int32_t *nbd, const void *g,
                      const void *r_in, double theta, int col, int head, void *xhat_out, int32_t *iword,
                      int32_t *info) = 0;
  virtual int r_lnsrlb(void *x, const void *l, const void *u, const int32_t *nbd, const void *g, double f,
                       double *sc, int32_t *ic, char *task, char *csave, int32_t *isave2, double *dsave13) = 0;
  virtual int r_matupd(const void *g, double stp, double dr, double dtd, int32_t *ip, double *theta_out) = 0;
  virtual int sync() = 0;
  // communicators (capi.hip): an initialised RCCL communicator / a host reducer for this context
  virtual int attach_rccl(ncclComm_t comm, int rank, int nranks) = 0;
  virtual ncclComm_t rccl_comm() const = 0;
  virtual int attach_host(lbfgsb_allreduce_fn ar, lbfgsb_allgather_fn ag, void *user, int rank,
                          int nranks) = 0;
  virtual void path_counts(int64_t &closed_form, int64_t &three_pass) const = 0;
  virtual int set_option(const char *name, double value) = 0;  // lbfgsb_hip_set_option
  virtual const void *prev_iterate() const = 0;  // t: the reference's wa(3n+2mn+11m^2+1 : +n)

  // host-entry staging (setulb_host)
  void *hx = nullptr, *hg = nullptr, *hl = nullptr, *hu = nullptr;
  int32_t *hnbd = nullptr;
  // ... what a call did to the vectors the host form mirrors back: x = t, g = r restored (:568-569, :736-737);
  // a line-search set-up ran (t = x, r = g written, :2235-2236) -- only then g / the t slot of wa travel D2H
  bool restored_xg = false;
  int64_t n_ls_setup = 0;
  // caller arrays of the host form pinned for the run (hipHostRegister at START: x, g, the t slot of wa), so
  // that the per-call copies are DMA transfers queued on the context's stream instead of staged pageable copies
  struct HostReg {
    void *p = nullptr;
    size_t bytes = 0;
  } host_reg[3];
  void host_register(int k, void *p, size_t bytes) {
    if (!p || bytes < ((size_t)1 << 20)) return;  // (small arrays: pinning costs more than it saves)
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess)
      host_reg[k].p = p, host_reg[k].bytes = bytes;
    else
      (void)hipGetLastError();  // (already registered by the caller, stack memory, ...: pageable copies then)
  }
  void host_unregister_if_not(int k, const void *p) {
    HostReg &r = host_reg[k];
    if (r.p && r.p != p) {
      if (hipHostUnregister(r.p) != hipSuccess) (void)hipGetLastError();
      r.p = nullptr, r.bytes = 0;
    }
  }
  void host_unregister_all() {
    for (HostReg &r : host_reg) {
      if (r.p && hipHostUnregister(r.p) != hipSuccess) (void)hipGetLastError();
      r.p = nullptr, r.bytes = 0;
    }
  }
  std::string itfile_name = "iterate.dat";
  int64_t n = 0, nglob = 0, row0 = 0;
  int m = 0, flags = 0, device = 0;
  int rank = 0, nranks = 1;
  int64_t nsync = 0, nfullsort = 0;
  int64_t ntiesplit = 0;  // walks that ended inside a group of equal breakpoints
  int64_t nrefresh = 0;   // memory refreshes of the current run (lbfgsb_hip_refresh_count)
  int64_t ngcp_clamped = 0;  // closed-form GCPs declined because the f2 clamp would have acted
  int64_t nspecwin = 0;   // walks served by the candidates the update pass handed over
  double t_wait = 0.0;  // seconds the host spent blocked in hipStreamSynchronize
  // host time between the landing of a trial point's sums and the launch of the next storing pass
  // (dcsrch, matupd, formt, the walk, formk's assembly and factorisations, the closed form): the
  // stretch of an iteration in which the device has nothing to do
  double t_mid = 0.0, t_mid0 = 0.0;
  int64_t n_mid = 0;
  // the same stretch cut at its milestones (LBFGSB_DEBUG prints the averages when the context goes):
  // 0 line search + return to the caller | 1 caller (NEW_X -> re-entry) | 2 termination tests, matupd, formt |
  // 3 cauchy (host walk; window syncs if any) + freev | 4 formk assembly / factorisations, closed form, triangular solves
  double t_seg[5] = {0, 0, 0, 0, 0}, t_mark = 0.0;
  void seg(int k) {
    if (t_mid0 <= 0.0) return;
    const double t = now_s();
    t_seg[k] += t - t_mark, t_mark = t;
  }
  int64_t ncoll = 0, coll_bytes = 0;  // collectives issued / bytes THIS rank contributed to them
  virtual int uniform_mask() const = 0;  // lbfgsb_hip_uniform_bounds
  // two device copies of (l, u, nbd) compared bit for bit -> number of rows that differ (host-pointer form)
  virtual int bounds_same(const void *l0, const void *u0, const int32_t *nb0, const void *l1, const void *u1,
                          const int32_t *nb1, double *ndiff) = 0;
  virtual int64_t freev_skipped() const = 0;
  virtual int collective_time(int reps, double *median_us, double *min_us) = 0;  // lbfgsb_hip_collective_time
  virtual int f_device(const double *d_f) = 0;                                   // lbfgsb_hip_f_device
  hipEvent_t return_ev = nullptr;                                                // lbfgsb_hip_return_event
  virtual void compact_stats(int64_t &packs, int64_t &unpacks, int &packed, int &eligible) const = 0;
  virtual int64_t skip_scans_reused() const = 0;
  virtual void defer_counts(int64_t &deferred, int64_t &reissued) const = 0;
  // a built-in objective whose value is still on the device (d_res[0], to be scaled by f_scale):
  // the next setulb_dev call fetches it together with the sums of its own first pass
  bool f_pending = false;
  double f_scale = 1.0;
  // in-run clocks of the three passes over W (hipEvents on the solver's stream around each
  // launch, read at the next host sync): 0 cmprlb_wtv, 1 update_scan, 2 subsm_update.  A small ring
  // of event pairs per pass: a pass launched again before the previous reading was collected takes
  // the next pair (every launch is counted; only a ring that is full drops its oldest reading, and
  // says so in clk_dropped)
  static constexpr int CLK_RING = 4;
  bool clock_on = false;
  hipEvent_t clk_ev[3][CLK_RING][2] = {};
  bool clk_pending[3][CLK_RING] = {};
  int clk_cur[3] = {0, 0, 0};
  double clk_ms[3] = {0.0, 0.0, 0.0};
  int64_t clk_n[3] = {0, 0, 0};
  int64_t clk_dropped = 0;
  hipStream_t clk_stream = nullptr;
  hipEvent_t order_ev = nullptr;  // lbfgsb_hip_wait_stream
  void clk_begin(int k) {
    if (!clock_on) return;
    clk_collect();
    int s = clk_cur[k];
    for (int tries = 0; tries < CLK_RING && clk_pending[k][s]; ++tries) s = (s + 1) % CLK_RING;
    if (clk_pending[k][s]) clk_dropped++, clk_pending[k][s] = false;
    clk_cur[k] = s;
    if (!clk_ev[k][s][0]) {
      (void)hipEventCreate(&clk_ev[k][s][0]);
      (void)hipEventCreate(&clk_ev[k][s][1]);
    }
    (void)hipEventRecord(clk_ev[k][s][0], clk_stream);
  }
  void clk_end(int k) {
    if (!clock_on) return;
    const int s = clk_cur[k];
    (void)hipEventRecord(clk_ev[k][s][1], clk_stream);
    clk_pending[k][s] = true;
    clk_cur[k] = (s + 1) % CLK_RING;
  }
  void clk_collect() {  // call after a stream sync (or when the events are known complete)
    for (int k = 0; k < 3; ++k)
      for (int s = 0; s < CLK_RING; ++s) {
        if (!clk_pending[k][s]) continue;
        if (hipEventQuery(clk_ev[k][s][1]) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, clk_ev[k][s][0], clk_ev[k][s][1]) == hipSuccess) {
          clk_ms[k] += ms;
          clk_n[k]++;
        }
        clk_pending[k][s] = false;
      }
  }
  lbk::Queue q{};
};


// the two instantiations live in solver.hip
lbfgsb_hip_ctx *lbfgsb_make_solver(int64_t n_local, int64_t n_global, int64_t row0, int m, int flags,
                                   int device, void *stream, int *rc);
