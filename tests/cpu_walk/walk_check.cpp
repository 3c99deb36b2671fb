// walk_check.cpp -- the library's HOST code of the generalized Cauchy point, compiled for the CPU and run under
// AddressSanitizer + UndefinedBehaviorSanitizer against the oracle's lbo_cauchy (reference src/lbfgsb.f90:1157-1532,
// hpsolb :2079-2157).  TEST INFRASTRUCTURE: built and run by tests/test_walk_cpu.py (pytest -m "not gpu").
//
// GPU sanitizers are not available on the pool, and the buffer-heavy part of this phase is host code anyway:
// the exact replay of the breakpoint walk (solver_walk.inl: cauchy(), walk_raw_col0), the breakpoint provider
// (solver_provider.inl: window_fetch, refill with its chunk sizes and prefetch, exchange / exchange_merged with
// the "safe to consume" bookkeeping over ranks, exact_init / refill_exact = the reference's heap order with 32-
// and 64-bit row numbers), the list of rows a walk fixes (row * 2 + bound), fetch()'s reduction over ranks, and
// import_state / export_state.  None of that is restated here: this file #includes the product's own
// solver.hip AS C++ (it is plain host code over the launch interface of kernels.hpp) and supplies
//   * a host stand-in for the few HIP runtime entry points it calls ("device" memory is malloc'ed host memory,
//     a stream is synchronous) -- so ASan also sees every access the host makes to a "device" buffer, and
//   * CPU twins of the KERNELS the Cauchy phase launches (scan, window compaction, sorts, record gathers, the
//     device merge of rank chunks, fix / finish, the Cauchy point as a vector), written from the kernels'
//     contracts in k_cauchy.hip / k_sort.hip.  A kernel that has no twin here is a null symbol: calling it
//     stops the run, which is how the harness says that a case left the phase under test.
// Several ranks are host threads, one context each, through the host-callback communicator (host merge of the
// rank chunks) or a stand-in for the RCCL entry points (ncclAllGather = a barrier + memcpy: the device-merge route).
//
// Cases: states of the ORACLE's own trajectories (lbo_setulb on random bounded problems: all four bound types,
// fixed variables, lattice data with many equal breakpoints, first iterations with ~n segments), each NEW_X
// state fed to lbo_cauchy and to the library's cauchy() with the same x, g, W, Sy, Wt, theta, iwhere; compared:
// iwhere and nseg exactly, xcp / p / c / wbp / v to 1e-10 / 1e-9, info.
#include "../../lbfgsb_amd/csrc/solver.hip"

#include <atomic>
#include <random>
#include <set>
#include <thread>

extern "C" {
#include "../../oracle/lbfgsb_oracle.h"
}

// ============================================================ HIP runtime stand-in (host memory, synchronous)
extern "C" {
hipError_t hipMalloc(void **p, size_t bytes) {
  *p = std::malloc(bytes ? bytes : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int) {
  *p = std::calloc(bytes ? bytes : 1, 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned int) {
  *dev = host;
  return hipSuccess;
}
hipError_t hipHostRegister(void *, size_t, unsigned int) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) {
  std::memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t) {
  std::memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) std::memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t) {
  std::memset(dst, v, bytes);
  return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t bytes) {
  std::memset(dst, v, bytes);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) {
  *s = reinterpret_cast<hipStream_t>(new int(0));
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  delete reinterpret_cast<int *>(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) {
  *e = reinterpret_cast<hipEvent_t>(new int(0));
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete reinterpret_cast<int *>(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
  *ms = 0.f;
  return hipSuccess;
}
hipError_t hipGetDeviceCount(int *c) {
  *c = 1;
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "host stand-in"; }
hipError_t hipMemGetInfo(size_t *fr, size_t *tot) {
  *fr = *tot = (size_t)1 << 34;
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *at, const void *p) {
  std::memset(at, 0, sizeof *at);
  at->type = hipMemoryTypeDevice;
  at->devicePointer = const_cast<void *>(p);
  return hipSuccess;
}
}

// ============================================================ CPU twins of the kernels of the Cauchy phase
namespace lbk {
namespace {
inline uint64_t key_of_h(double t) {
  uint64_t b;
  std::memcpy(&b, &t, 8);
  return b;
}
inline bool after_cursor_h(double t, int64_t gi, double lo_t, int64_t lo_i) { return t > lo_t || (t == lo_t && gi > lo_i); }
const double INF = std::numeric_limits<double>::infinity();
// cauchy :1284-1291 -- iwhere of one row after the n-loop's update
inline int scan_iw(int iw, int nb, double x, double l, double u, double g) {
  if (iw != 3 && iw != -1) {
    const double neggi = -g;
    double tl = 0.0, tu = 0.0;
    if (nb <= 2) tl = x - l;
    if (nb >= 2) tu = u - x;
    const bool xlower = nb <= 2 && tl <= 0.0, xupper = nb >= 2 && tu <= 0.0;
    iw = 0;
    if (xlower) {
      if (neggi <= 0.0) iw = 1;
    } else if (xupper) {
      if (neggi >= 0.0) iw = 2;
    } else if (std::fabs(neggi) <= 0.0) {
      iw = -3;
    }
  }
  return iw;
}
template <typename T>
inline double brk_time_h(double x, double l, double u, int nb, double g, int iw) {
  if (iw != 0 && iw != -1) return -1.0;
  const double neggi = -g;
  double tb = INF;
  if (nb <= 2 && nb != 0 && neggi < 0.0)
    tb = (x - l) / (-neggi);
  else if (nb >= 2 && neggi > 0.0)
    tb = (u - x) / neggi;
  return (double)(T)tb;
}
template <typename T>
inline double xcp_row_h(double x, double g, int iw, double l, double u, double tsum) {
  if (tsum == 0.0) return x;
  if (iw == 1) return (x > l ? l : x) + tsum * 0.0;
  if (iw == 2) return (x < u ? u : x) + tsum * 0.0;
  if (iw == 0 || iw == -1) return tsum != 0.0 ? (double)(T)(x + tsum * (-g)) : x;
  return x + tsum * 0.0;
}
template <typename T>
inline double pend_y_h(double g, double r) { return (double)(T)(g - r); }
template <typename T>
inline double pend_sx_h(double t, double x, Pend pe) {
  if (pe.impl) return (double)(T)(x - t);
  return pe.stp != 1.0 ? (double)(T)(pe.stp * t) : t;
}
}  // namespace

int maxc_for(int col) { return col <= 5 ? 5 : col <= 10 ? 10 : col <= 20 ? 20 : 32; }
void launch_finalize(Queue &, int, int, int, int) {}
void finalize_flush(Queue &) {}
void launch_publish(Queue &q, const double *src, double *dst, int count, unsigned long long seq,
                    unsigned long long *flag) {
  std::memcpy(dst, src, (size_t)count * sizeof(double));
  __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
  q.launches++;
}
// the tile-local free-row layout of W (k_layout.hip): this harness never turns the option on, so the bits stay "every
// row of [0, n)" = natural order; a re-sort request would be a bug of the host code under test
void launch_lmask_ones(Queue &q, int64_t n, uint64_t *lmask) {
  const int64_t nwords = ((n + CW_TILE - 1) / CW_TILE) * (CW_TILE / 64);
  for (int64_t w = 0; w < nwords; ++w) {
    const int64_t r0 = w * 64;
    lmask[w] = r0 + 64 <= n ? ~0ull : (r0 < n ? (1ull << (n - r0)) - 1ull : 0ull);
  }
  q.launches++;
}
template <typename T>
void launch_w_relayout(Queue &, int64_t, const iw_t *, uint64_t *, WStore<T>, int, int) {
  std::fprintf(stderr, "walk_check: launch_w_relayout called (the option is off in this harness)\n");
  std::abort();
}
template void launch_w_relayout<double>(Queue &, int64_t, const iw_t *, uint64_t *, WStore<double>, int, int);
template void launch_w_relayout<float>(Queue &, int64_t, const iw_t *, uint64_t *, WStore<float>, int, int);
void launch_nbd_pack(Queue &q, int64_t n, const int32_t *nbd, nb_t *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (nb_t)nbd[i];
  q.launches++;
}
size_t sort_pairs_temp_bytes(size_t count) { return 64 + count; }
void launch_sort_by_idx(Queue &q, void *, size_t, const uint32_t *idx_in, uint32_t *idx_out, const uint64_t *keys_in,
                        uint64_t *keys_out, size_t count) {
  std::vector<size_t> o(count);
  for (size_t k = 0; k < count; ++k) o[k] = k;
  std::stable_sort(o.begin(), o.end(), [&](size_t a, size_t b) { return idx_in[a] < idx_in[b]; });
  for (size_t k = 0; k < count; ++k) idx_out[k] = idx_in[o[k]], keys_out[k] = keys_in[o[k]];
  q.launches++;
}
void launch_sort_pairs(Queue &q, void *, size_t, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *idx_in,
                       uint32_t *idx_out, size_t count) {
  std::vector<size_t> o(count);
  for (size_t k = 0; k < count; ++k) o[k] = k;
  std::stable_sort(o.begin(), o.end(), [&](size_t a, size_t b) { return keys_in[a] < keys_in[b]; });
  for (size_t k = 0; k < count; ++k) idx_out[k] = idx_in[o[k]], keys_out[k] = keys_in[o[k]];
  q.launches++;
}
void launch_merge_chunks(Queue &q, int nranks, uint32_t chunk, int recl, size_t stride, const double *all, uint64_t *keys0,
                         uint64_t *keys1, uint32_t *vals0, uint32_t *vals1, void *, size_t, double *out) {
  const size_t S = (size_t)nranks * chunk;
  for (size_t s = 0; s < S; ++s) {
    const int rk = (int)(s / chunk);
    const uint32_t k = (uint32_t)(s % chunk);
    const double *base = all + (size_t)rk * stride;
    const uint32_t lr = (uint32_t)base[0];
    keys0[s] = k < lr ? key_of_h(base[2 + (size_t)k * recl]) : ~0ull;
    vals0[s] = (uint32_t)s;
    if (k == 0) {
      out[4 * rk + 0] = base[0], out[4 * rk + 1] = base[1];
      out[4 * rk + 2] = lr ? base[2 + (size_t)(lr - 1) * recl] : 0.0;
      out[4 * rk + 3] = lr ? base[2 + (size_t)(lr - 1) * recl + 1] : 0.0;
    }
  }
  launch_sort_pairs(q, nullptr, 0, keys0, keys1, vals0, vals1, S);
  double *recs = out + 4 * (size_t)nranks;
  unsigned char *rb = reinterpret_cast<unsigned char *>(recs + S * (size_t)recl);
  for (size_t p = 0; p < S; ++p) {
    if (keys1[p] == ~0ull) continue;
    const uint32_t s = vals1[p];
    const int rk = (int)(s / chunk);
    const uint32_t k = s % chunk;
    for (int f = 0; f < recl; ++f) recs[p * recl + f] = all[(size_t)rk * stride + 2 + (size_t)k * recl + f];
    rb[p] = (unsigned char)rk;
  }
}
void launch_cauchy_fix(Queue &q, const int64_t *list, int count, int64_t row0, int64_t n, iw_t *iwhere) {
  for (int k = 0; k < count; ++k) {
    const int64_t gi = list[k] >> 1;
    if (gi >= row0 && gi < row0 + n) iwhere[gi - row0] = (list[k] & 1) ? 2 : 1;
  }
  q.launches++;
}

template <typename T>
void launch_cauchy_scan(Queue &q, int64_t n, const T *x, const T *l, const T *u, const int32_t *nbd, const T *g,
                        iw_t *iwhere, T *tbrk, WStore<T> w, int head, int col) {
  const int MC = col == 0 ? 0 : maxc_for(col);
  double *res = q.d_res + q.res_off;
  for (int k = 0; k < 2 * MC + 4; ++k) res[k] = 0.0;
  double bkmin = INF;
  for (int64_t i = 0; i < n; ++i) {
    const double xv = (double)x[i], lv = (double)l[i], uv = (double)u[i], gv = (double)g[i];
    const int nb = nbd[i];
    const int iw = scan_iw(iwhere[i], nb, xv, lv, uv, gv);
    const double neggi = -gv;
    double tb, ng;
    if (iw != 0 && iw != -1) {
      tb = -1.0, ng = 0.0;
    } else {
      ng = neggi;
      res[2 * MC] = res[2 * MC] - neggi * neggi;
      const double tl = nb <= 2 ? xv - lv : 0.0, tu = nb >= 2 ? uv - xv : 0.0;
      if (nb <= 2 && nb != 0 && neggi < 0.0) {
        tb = tl / (-neggi), res[2 * MC + 1] += 1.0, bkmin = std::fmin(bkmin, tb);
      } else if (nb >= 2 && neggi > 0.0) {
        tb = tu / neggi, res[2 * MC + 1] += 1.0, bkmin = std::fmin(bkmin, tb);
      } else {
        tb = INF, res[2 * MC + 2] += 1.0;
        if (std::fabs(neggi) > 0.0) res[2 * MC + 3] += 1.0;
      }
    }
    for (int j = 0; j < col; ++j) {
      const int64_t off = (int64_t)((head - 1 + j) % w.m) * w.ld + i;
      res[j] += (double)w.wy[off] * ng;
      res[MC + j] += (double)w.ws[off] * ng;
    }
    iwhere[i] = (iw_t)iw;
    tbrk[i] = (T)tb;
  }
  res[2 * MC + 4] = bkmin;
  q.launches++;
}
template <typename T>
void launch_tbrk_fill(Queue &q, int64_t n, const T *x, const T *l, const T *u, const int32_t *nbd, const T *g,
                      const iw_t *iwhere, T *tbrk) {
  for (int64_t i = 0; i < n; ++i)
    tbrk[i] = (T)brk_time_h<T>((double)x[i], (double)l[i], (double)u[i], nbd[i], (double)g[i], iwhere[i]);
  q.launches++;
}
template <typename T>
void launch_cauchy_window(Queue &q, int64_t n, int64_t row0, const T *tbrk, double lo_t, int64_t lo_i, double hi_t,
                          uint64_t *keys, uint32_t *idx, uint32_t cap, uint32_t *d_count) {
  uint32_t pos = 0;
  // (appended from the BACK: the kernel's order is not defined, every consumer sorts)
  for (int64_t i = n - 1; i >= 0; --i) {
    const double t = (double)tbrk[i];
    if (t >= 0.0 && t <= hi_t && after_cursor_h(t, row0 + i, lo_t, lo_i)) {
      if (pos < cap) keys[pos] = key_of_h(t), idx[pos] = (uint32_t)i;
      ++pos;
    }
  }
  *d_count = pos;
  q.launches++;
}
template <typename T>
void launch_cauchy_window_fly(Queue &q, int64_t n, int64_t row0, const T *x, const T *l, const T *u, const nb_t *nbd,
                              const T *g, const iw_t *iwhere, double lo_t, int64_t lo_i, double hi_t, uint64_t *keys,
                              uint32_t *idx, uint32_t cap, uint32_t *d_count, int ub) {
  uint32_t pos = 0;
  for (int64_t i = n - 1; i >= 0; --i) {
    unsigned code = (unsigned)(unsigned char)((ub & 4) ? nbd[0] : nbd[i]);
    double lv = (double)((ub & 1) ? l[0] : l[i]), uv = (double)((ub & 2) ? u[0] : u[i]);
    int nb = (int)(signed char)code;
    if (ub & UB_DICT) lv = (double)l[(code >> 2) & 7u], uv = (double)u[code >> 5], nb = (int)(code & 3u);
    const double t = brk_time_h<T>((double)x[i], lv, uv, nb, (double)g[i], iwhere[i]);
    if (t >= 0.0 && t <= hi_t && after_cursor_h(t, row0 + i, lo_t, lo_i)) {
      if (pos < cap) keys[pos] = key_of_h(t), idx[pos] = (uint32_t)i;
      ++pos;
    }
  }
  *d_count = pos;
  q.launches++;
}
template <typename T>
void launch_cauchy_allkeys(Queue &q, int64_t n, int64_t row0, const T *tbrk, double lo_t, int64_t lo_i, uint64_t *keys,
                           uint32_t *idx) {
  for (int64_t i = 0; i < n; ++i) {
    const double t = (double)tbrk[i];
    const bool pred = t >= 0.0 && t < INF && after_cursor_h(t, row0 + i, lo_t, lo_i);
    keys[i] = pred ? key_of_h(t) : ~0ull;
    idx[i] = (uint32_t)i;
  }
  q.launches++;
}
template <typename T>
static void gather_records(const uint32_t *idx, const uint64_t *keys, uint32_t cnt, int64_t row0, const T *x, const T *l,
                           const T *u, const T *g, WStore<T> w, int head, int col, const T *pr, const T *pd, Pend pe,
                           double *rec) {
  const int rl = 2 * col + 4;
  for (uint32_t k = 0; k < cnt; ++k) {
    const int64_t i = idx[k];
    double *o = rec + (size_t)k * rl;
    std::memcpy(&o[0], &keys[k], 8);
    o[1] = (double)(row0 + i);
    const double d = -(double)g[i];
    o[2] = d;
    o[3] = d > 0.0 ? (double)u[i] - (double)x[i] : (double)l[i] - (double)x[i];
    for (int j = 0; j < col; ++j) {
      const int64_t off = (int64_t)((head - 1 + j) % w.m) * w.ld + i;
      o[4 + j] = (pe.on && j == col - 1) ? pend_y_h<T>((double)g[i], (double)pr[i]) : (double)w.wy[off];
      o[4 + col + j] = (pe.on && j == col - 1) ? pend_sx_h<T>((double)pd[i], (double)x[i], pe) : (double)w.ws[off];
    }
  }
}
template <typename T>
void launch_cauchy_gather(Queue &q, const uint32_t *idx, const uint64_t *keys, uint32_t cnt, int64_t row0, const T *x,
                          const T *l, const T *u, const T *g, WStore<T> w, int head, int col, const T *pr, const T *pd,
                          Pend pe, double *rec) {
  gather_records<T>(idx, keys, cnt, row0, x, l, u, g, w, head, col, pr, pd, pe, rec);
  q.launches++;
}
template <typename T>
void launch_cauchy_gather_dyn(Queue &q, const uint32_t *idx, const uint64_t *keys, const uint32_t *d_count, uint32_t cap,
                              int64_t row0, const T *x, const T *l, const T *u, const T *g, WStore<T> w, int head, int col,
                              const T *pr, const T *pd, Pend pe, double *msg) {
  const uint32_t total = *d_count, cnt = total < cap ? total : cap;
  msg[0] = (double)total, msg[1] = 0.0;
  gather_records<T>(idx, keys, cnt, row0, x, l, u, g, w, head, col, pr, pd, pe, msg + 2);
  q.launches++;
}
template <typename T>
void launch_cauchy_finish(Queue &q, int64_t n, int64_t row0, const T *x, const T *l, const T *u, const T *g, const T *tbrk,
                          iw_t *iwhere, T *xcp, double tsum, double last_t, int64_t last_i, int count) {
  const double poison = tsum * 0.0;
  double cnt = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double xv = (double)x[i], gv = (double)g[i], tb = (double)tbrk[i];
    const bool done = tb >= 0.0 && (tb < last_t || (tb == last_t && (row0 + i) <= last_i));
    if (done) cnt += 1.0;
    double out = xv + poison;
    if (tb >= 0.0) {
      const double d = -gv;
      if (done) {
        if (d > 0.0)
          out = (double)u[i] + poison, iwhere[i] = 2;
        else
          out = (double)l[i] + poison, iwhere[i] = 1;
      } else if (tsum != 0.0) {
        out = xv + tsum * d;
      } else {
        out = xv;
      }
    }
    xcp[i] = (T)out;
  }
  if (count) q.d_res[q.res_off] = cnt;
  q.launches++;
}
template <typename T>
void launch_xcp_fill(Queue &q, int64_t n, const T *x, const T *g, const T *l, const T *u, const iw_t *iwhere, double tsum,
                     T *dst) {
  for (int64_t i = 0; i < n; ++i)
    dst[i] = (T)xcp_row_h<T>((double)x[i], (double)g[i], iwhere[i], (double)l[i], (double)u[i], tsum);
  q.launches++;
}
template <typename T>
void launch_dz_materialise(Queue &q, int64_t n, const T *x, const T *t, T *d, T *z) {
  for (int64_t i = 0; i < n; ++i) {
    d[i] = (T)((double)x[i] - (double)t[i]);
    if (z) z[i] = x[i];
  }
  q.launches++;
}

// ---- task 'START': errclb (+ the uniformity flags), the bound dictionary's probe and pack, active ----
template <typename T>
void launch_errclb(Queue &q, int64_t n, int64_t row0, const T *l, const T *u, const int32_t *nbd) {
  double *res = q.d_res + q.res_off;
  for (int k = 0; k < 5; ++k) res[k] = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double gi = (double)(row0 + i + 1);
    if (nbd[i] < 0 || nbd[i] > 3) res[0] = std::fmax(res[0], gi);
    if (nbd[i] == 2 && (double)l[i] > (double)u[i]) res[1] = std::fmax(res[1], gi);
    if (std::memcmp(&l[i], &l[0], sizeof(T)) != 0) res[2] = 1.0;
    if (std::memcmp(&u[i], &u[0], sizeof(T)) != 0) res[3] = 1.0;
    if (nbd[i] != nbd[0]) res[4] = 1.0;
  }
  q.launches++;
}
static int dict_find_h(const BoundTables &tb, int which, double v) {
  const int cnt = which ? tb.nu : tb.nl;
  for (int j = 0; j < cnt && j < 8; ++j)
    if (std::memcmp(which ? &tb.u[j] : &tb.l[j], &v, 8) == 0) return j;
  return -1;
}
template <typename T>
void launch_dict_probe(Queue &q, int64_t n, const T *l, const T *u, const BoundTables &tb) {
  double *res = q.d_res + q.res_off;
  res[0] = res[1] = 0.0, res[2] = res[3] = INF;
  for (int64_t i = 0; i < n; ++i) {
    if (dict_find_h(tb, 0, (double)l[i]) < 0) res[0] += 1.0, res[2] = std::fmin(res[2], (double)l[i]);
    if (dict_find_h(tb, 1, (double)u[i]) < 0) res[1] += 1.0, res[3] = std::fmin(res[3], (double)u[i]);
  }
  q.launches++;
}
template <typename T>
void launch_nbd_pack_dict(Queue &q, int64_t n, const int32_t *nbd, const T *l, const T *u, const BoundTables &tb,
                          nb_t *out) {
  for (int64_t i = 0; i < n; ++i) {
    const int jl = dict_find_h(tb, 0, (double)l[i]), ju = dict_find_h(tb, 1, (double)u[i]);
    out[i] = (nb_t)(unsigned char)((unsigned)(nbd[i] & 3) | ((unsigned)(jl < 0 ? 0 : jl) << 2) |
                                   ((unsigned)(ju < 0 ? 0 : ju) << 5));
  }
  q.launches++;
}
template <typename T>
void launch_active(Queue &q, int64_t n, T *x, const T *l, const T *u, const int32_t *nbd, iw_t *iwhere,
                   int8_t *wasfree) {
  double *res = q.d_res + q.res_off;
  for (int k = 0; k < 4; ++k) res[k] = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const int nb = nbd[i];
    double xv = (double)x[i];
    const double lv = (double)l[i], uv = (double)u[i];
    if (nb > 0) {
      if (nb <= 2 && xv <= lv) {
        if (xv < lv) res[0] += 1.0, xv = lv;
        res[3] += 1.0;
      } else if (nb >= 2 && xv >= uv) {
        if (xv > uv) res[0] += 1.0, xv = uv;
        res[3] += 1.0;
      }
    }
    if (nb != 2) res[2] += 1.0;
    if (nb == 0) {
      iwhere[i] = -1;
    } else {
      res[1] += 1.0;
      iwhere[i] = (nb == 2 && uv - lv <= 0.0) ? 3 : 0;
    }
    wasfree[i] = 1;
    x[i] = (T)xv;
  }
  q.launches++;
}
#define INST(T)                                                                                                          \
  template void launch_cauchy_scan<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *,    \
                                      iw_t *, T *, WStore<T>, int, int);                                                \
  template void launch_tbrk_fill<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *,      \
                                    const iw_t *, T *);                                                                 \
  template void launch_cauchy_window<T>(Queue &, int64_t, int64_t, const T *, double, int64_t, double, uint64_t *,      \
                                        uint32_t *, uint32_t, uint32_t *);                                              \
  template void launch_cauchy_window_fly<T>(Queue &, int64_t, int64_t, const T *, const T *, const T *, const nb_t *,   \
                                            const T *, const iw_t *, double, int64_t, double, uint64_t *, uint32_t *,   \
                                            uint32_t, uint32_t *, int);                                                 \
  template void launch_cauchy_allkeys<T>(Queue &, int64_t, int64_t, const T *, double, int64_t, uint64_t *, uint32_t *); \
  template void launch_cauchy_gather<T>(Queue &, const uint32_t *, const uint64_t *, uint32_t, int64_t, const T *,      \
                                        const T *, const T *, const T *, WStore<T>, int, int, const T *, const T *,     \
                                        Pend, double *);                                                                \
  template void launch_cauchy_gather_dyn<T>(Queue &, const uint32_t *, const uint64_t *, const uint32_t *, uint32_t,    \
                                            int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int,   \
                                            const T *, const T *, Pend, double *);                                      \
  template void launch_cauchy_finish<T>(Queue &, int64_t, int64_t, const T *, const T *, const T *, const T *,          \
                                        const T *, iw_t *, T *, double, double, int64_t, int);                          \
  template void launch_xcp_fill<T>(Queue &, int64_t, const T *, const T *, const T *, const T *, const iw_t *, double,  \
                                   T *);                                                                                \
  template void launch_dz_materialise<T>(Queue &, int64_t, const T *, const T *, T *, T *);                            \
  template void launch_errclb<T>(Queue &, int64_t, int64_t, const T *, const T *, const int32_t *);                    \
  template void launch_dict_probe<T>(Queue &, int64_t, const T *, const T *, const BoundTables &);                     \
  template void launch_nbd_pack_dict<T>(Queue &, int64_t, const int32_t *, const T *, const T *, const BoundTables &,  \
                                        nb_t *);                                                                        \
  template void launch_active<T>(Queue &, int64_t, T *, const T *, const T *, const int32_t *, iw_t *, int8_t *);
INST(double)
INST(float)
#undef INST
}  // namespace lbk

// ============================================================ several ranks = host threads
struct Barrier {
  std::mutex mu;
  std::condition_variable cv;
  int n = 1, arrived = 0;
  unsigned long gen = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long g = gen;
    if (++arrived == n) {
      arrived = 0, ++gen;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};
struct World {  // what the rank threads of one case share
  int nranks = 1;
  Barrier bar;
  std::vector<const void *> src;
  std::vector<double> red;
};
struct RankComm {
  World *w;
  int rank;
};
// all-gather `bytes` from every rank into out (rank-major): barrier, copy, barrier
static void world_allgather(RankComm *c, const void *in, void *out, size_t bytes) {
  c->w->src[c->rank] = in;
  c->w->bar.wait();
  for (int r = 0; r < c->w->nranks; ++r) std::memcpy((char *)out + (size_t)r * bytes, c->w->src[r], bytes);
  c->w->bar.wait();
}
static int cb_allgather(void *user, const void *in, void *out, int64_t bytes) {
  world_allgather((RankComm *)user, in, out, (size_t)bytes);
  return 0;
}
static int cb_allreduce(void *user, double *buf, int nsum, int nmin, int nmax) {
  RankComm *c = (RankComm *)user;
  const int k = nsum + nmin + nmax;
  std::vector<double> all((size_t)k * c->w->nranks);
  world_allgather(c, buf, all.data(), (size_t)k * 8);
  for (int j = 0; j < k; ++j) {
    double v = all[j];
    for (int r = 1; r < c->w->nranks; ++r) {
      const double x = all[(size_t)r * k + j];
      v = j < nsum ? v + x : (j < nsum + nmin ? std::fmin(v, x) : std::fmax(v, x));
    }
    buf[j] = v;
  }
  return 0;
}
// stand-in for the RCCL entry points the library calls (ncclComm_t = RankComm *): the communicator route,
// i.e. the all-gathers on "device" buffers and the DEVICE merge of the rank chunks (exchange_merged)
static ncclResult_t fake_allgather(const void *send, void *recv, size_t count, ncclDataType_t, ncclComm_t comm, hipStream_t) {
  world_allgather(reinterpret_cast<RankComm *>(comm), send, recv, count * 8);
  return ncclSuccess;
}
static ncclResult_t fake_destroy(ncclComm_t) { return ncclSuccess; }

// ============================================================ the cases
static std::atomic<long> g_fullsorts{0}, g_tiesplits{0}, g_syncs{0}, g_collectives{0};
struct Case {
  int n, m, col, head;
  double theta, sbgnrm;
  std::vector<double> x, l, u, g, ws, wy, sy, wt;  // ws, wy: n x m column-major (ld = n)
  std::vector<int> nbd, iwhere;
};
struct Out {
  std::vector<int> iwhere;
  std::vector<double> xcp, pcwv;  // p, c, wbp, v: 4 x 2m
  int nseg = 0, info = 0;
};
static void run_oracle(const Case &c, Out &o) {
  const int n = c.n, m = c.m;
  o.iwhere = c.iwhere;
  o.xcp.assign(n, 0.0);
  o.pcwv.assign((size_t)8 * m, 0.0);
  std::vector<int> iorder(n);
  std::vector<double> t(n), d(n);
  lbo_cauchy(n, c.x.data(), c.l.data(), c.u.data(), c.nbd.data(), c.g.data(), iorder.data(), o.iwhere.data(), t.data(),
             d.data(), o.xcp.data(), m, c.wy.data(), c.ws.data(), c.sy.data(), c.wt.data(), c.theta, c.col, c.head,
             &o.pcwv[0], &o.pcwv[2 * m], &o.pcwv[4 * m], &o.pcwv[6 * m], &o.nseg, c.sbgnrm, &o.info,
             std::numeric_limits<double>::epsilon());
}
struct Opts {
  int nranks = 1;
  int comm_kind = 0;       // several ranks: 0 host callbacks (host merge), 1 communicator stand-in (device merge)
  bool exact_always = false;
  bool index_ties = false;
  bool via_door = true;    // single rank: import_state + the cauchy door + export_state
  int64_t row_base = 0;    // global number of this case's first row (2^31 / 2^32 neighbourhoods)
};
// one rank's part of a case through the library's own cauchy()
static int run_rank(const Case &c, const Opts &op, World *w, int rank, Out &o, std::string &err) {
  using S = Solver<double>;
  const int n = c.n, m = c.m;
  const int64_t lo = (int64_t)n * rank / op.nranks, hi = (int64_t)n * (rank + 1) / op.nranks, nl = hi - lo;
  const int64_t nglob = op.row_base + n;
  int rc = 0;
  int flags = op.index_ties ? LBFGSB_F_INDEX_TIES : 0;
  S *s = new S();
  rc = s->init(nl, nglob, op.row_base + lo, m, flags, 0, nullptr);
  if (rc) {
    err = "init: " + g_err;
    delete s;
    return rc;
  }
  RankComm rcomm{w, rank};
  if (op.nranks > 1) {
    if (op.comm_kind == 1)
      rc = s->attach_rccl(reinterpret_cast<ncclComm_t>(&rcomm), rank, op.nranks);
    else
      rc = s->attach_host(cb_allreduce, cb_allgather, &rcomm, rank, op.nranks);
  }
  if (!rc) rc = s->set_option("spin", 0.0);  // (results by copy + sync: the stand-in stream is synchronous)
  if (!rc && op.exact_always) rc = s->set_option("exact_always", 1.0);
  std::vector<int32_t> nbd(c.nbd.begin() + lo, c.nbd.begin() + hi);
  int nseg = 0, info = 0;
  std::vector<double> xcp(nl);
  std::vector<int32_t> iw_out(nl);
  if (!rc && op.nranks == 1 && op.via_door) {
    // the reference's wa / iwa layout in, the door, the layout out
    const int64_t wl = lbo_wa_len(n, m);
    int64_t off[13];
    lbo_wa_offsets(n, m, off);
    std::vector<double> wa((size_t)wl, 0.0);
    std::vector<int32_t> iwa((size_t)3 * n, 0), isave(44, 0);
    std::copy(c.ws.begin(), c.ws.end(), wa.begin() + off[0]);
    std::copy(c.wy.begin(), c.wy.end(), wa.begin() + off[1]);
    std::copy(c.sy.begin(), c.sy.end(), wa.begin() + off[2]);
    std::copy(c.wt.begin(), c.wt.end(), wa.begin() + off[4]);
    for (int i = 0; i < n; ++i) iwa[(size_t)n + i] = c.iwhere[i];
    rc = s->import_state(wa.data(), iwa.data(), isave.data());
    if (!rc)
      rc = s->r_cauchy(c.x.data(), c.l.data(), c.u.data(), nbd.data(), c.g.data(), c.theta, c.col, c.head, c.sbgnrm,
                       xcp.data(), &nseg, &info);
    if (!rc) rc = s->export_state(wa.data(), iwa.data());
    for (int i = 0; i < n; ++i) iw_out[i] = iwa[(size_t)n + i];
    o.pcwv.assign(wa.begin() + off[12], wa.begin() + off[12] + 8 * m);
  } else if (!rc) {
    // several ranks (the doors are single-rank): the members the iteration itself uses
    for (int j = 0; j < m; ++j) {
      std::memcpy(s->ws + (size_t)j * s->ld, c.ws.data() + (size_t)j * n + lo, (size_t)nl * 8);
      std::memcpy(s->wy + (size_t)j * s->ld, c.wy.data() + (size_t)j * n + lo, (size_t)nl * 8);
    }
    s->sy = c.sy, s->wt = c.wt;
    for (int64_t i = 0; i < nl; ++i) s->iwhere[i] = (lbk::iw_t)c.iwhere[lo + i];
    s->iw_dirty = 1.0;
    rc = s->door_ready(nullptr);
    if (rc == LBFGSB_E_STATE) rc = 0;  // ("single-rank": the state it resets is reset all the same)
    s->nbd8_src = nullptr;
    if (!rc) rc = s->ensure_nbd8(nbd.data());
    s->print_level = -1, s->quiet = rank != 0;
    if (!rc)
      rc = s->cauchy(c.x.data() + lo, c.l.data() + lo, c.u.data() + lo, nbd.data(), c.g.data() + lo, c.theta, c.col,
                     c.head, c.sbgnrm, std::numeric_limits<double>::epsilon(), nseg, info);
    if (!rc && info == 0) rc = s->ensure_z(c.x.data() + lo, c.l.data() + lo, c.u.data() + lo, c.g.data() + lo);
    if (!rc) {
      std::copy(s->z, s->z + nl, xcp.begin());
      for (int64_t i = 0; i < nl; ++i) iw_out[i] = s->iwhere[i];
      o.pcwv = s->wa8m;
    }
  }
  if (rc) err = "rank " + std::to_string(rank) + ": " + g_err;
  o.nseg = nseg, o.info = info;
  o.xcp.assign(xcp.begin(), xcp.end());
  o.iwhere.assign(iw_out.begin(), iw_out.end());
  if (op.nranks > 1 && op.comm_kind == 1) s->comm = nullptr;  // (not a real communicator: nothing to destroy)
  if (rank == 0) g_fullsorts += s->nfullsort, g_tiesplits += s->ntiesplit, g_syncs += s->nsync, g_collectives += s->ncoll;
  delete s;
  return rc;
}
static bool close_to(const double *a, const double *b, size_t k, double tol, double &worst) {
  double scale = 1.0, err = 0.0;
  for (size_t i = 0; i < k; ++i) scale = std::max(scale, std::fabs(b[i]));
  for (size_t i = 0; i < k; ++i) {
    const double e = (std::isnan(a[i]) && std::isnan(b[i])) ? 0.0 : std::fabs(a[i] - b[i]);
    if (!(e <= err)) err = e;
  }
  worst = err / scale;
  return err <= tol * scale;
}
// -> "" if the library's result equals the oracle's; "~..." for the one tolerated class of differences:
// THE STATIONARY POINT ON A BREAKPOINT.  On lattice data the walk can end where t_j + dtm equals the next
// breakpoint to the last bit (theta = 1 after the first iteration: t* = 1/theta = 1 is a lattice value).  Whether
// that breakpoint is crossed -- its rows fixed at their bounds, or left free AT their bounds -- then rests on the
// last bit of dtm = -f1/f2, i.e. on the rounding of f1 = -sum g_i^2, an n-term sum that several ranks (and every
// GPU reduction) add in another order than the reference's loop, and on the order in which a group of equal
// breakpoints was added up.  DESIGN.md section 7 lists it (reassociated sums; decisions inside their rounding
// noise).  Accepted only if: the Cauchy point itself agrees to 1e-10, every row whose iwhere differs has ONE
// common breakpoint time, that time is where the walk ended, and the segment counts differ by exactly those rows.
static std::string check_case(const Case &c, const Opts &op) {
  Out want;
  run_oracle(c, want);
  World w;
  w.nranks = op.nranks, w.bar.n = op.nranks, w.src.assign(op.nranks, nullptr);
  std::vector<Out> got(op.nranks);
  std::vector<std::string> errs(op.nranks);
  std::vector<int> rcs(op.nranks, 0);
  if (op.nranks == 1) {
    rcs[0] = run_rank(c, op, &w, 0, got[0], errs[0]);
  } else {
    std::vector<std::thread> th;
    for (int r = 0; r < op.nranks; ++r)
      th.emplace_back([&, r] { rcs[r] = run_rank(c, op, &w, r, got[r], errs[r]); });
    for (auto &t : th) t.join();
  }
  for (int r = 0; r < op.nranks; ++r)
    if (rcs[r]) return "library error " + std::to_string(rcs[r]) + " (" + errs[r] + ")";
  const int n = c.n, m = c.m;
  std::vector<int> iw;
  std::vector<double> xcp;
  for (int r = 0; r < op.nranks; ++r) {
    iw.insert(iw.end(), got[r].iwhere.begin(), got[r].iwhere.end());
    xcp.insert(xcp.end(), got[r].xcp.begin(), got[r].xcp.end());
    if (got[r].nseg != got[0].nseg || got[r].info != got[0].info) return "the ranks disagree on nseg / info";
    if (got[r].info != want.info)
      return "info " + std::to_string(got[r].info) + " vs " + std::to_string(want.info);
  }
  if (want.info != 0) return "";
  if ((int)iw.size() != n) return "row count";
  double worst = 0.0;
  // One rank: the twins add f1 = -sum g_i^2 and p = W'd in the reference's own order, the walk is an exact replay:
  // 1e-10 whatever n.  Several ranks: per-rank partial sums, added in rank order -- the walk ends where f1 crosses
  // zero, and f1 carries the rounding of its n-term starting value all the way (tsum moves by
  // ~ eps |f1(0)| / f2(end)); the 1e-10 bar of the GPU tests is set at n <= 2e4 and scales with the number of terms.
  // The conditioning, from the data: kappa = (sum of g_i^2 over the rows that move at all) / (the same over the
  // rows still moving at the end of the walk) ~ |f1(0)| / f2(end).
  double g2_all = 0.0, g2_end = 0.0;
  for (int i = 0; i < n; ++i) {
    const int iws = lbk::scan_iw(c.iwhere[i], c.nbd[i], c.x[i], c.l[i], c.u[i], c.g[i]);
    if (iws == 0 || iws == -1) {
      g2_all += c.g[i] * c.g[i];
      if (want.iwhere[i] == 0 || want.iwhere[i] == -1) g2_end += c.g[i] * c.g[i];
    }
  }
  const double kappa = g2_end > 0.0 ? g2_all / g2_end : 1.0;
  const double xtol = 1e-10 * (op.nranks > 1 ? std::max(1.0, std::max((double)n / 2.0e4, kappa / 100.0)) : 1.0);
  if (!close_to(xcp.data(), want.xcp.data(), n, xtol, worst)) {
    int at = 0;
    for (int i = 0; i < n; ++i)
      if (std::fabs(xcp[i] - want.xcp[i]) > std::fabs(xcp[at] - want.xcp[at])) at = i;
    char buf[200];
    std::snprintf(buf, sizeof buf, "xcp off by %.3g > %.3g, kappa %.3g (row %d: %.17g vs %.17g, iwhere %d vs %d; nseg %d vs %d)",
                  worst, xtol, kappa, at, xcp[at], want.xcp[at], iw[at], want.iwhere[at], got[0].nseg, want.nseg);
    return buf;
  }
  std::vector<int> diff;
  for (int i = 0; i < n; ++i)
    if (iw[i] != want.iwhere[i]) diff.push_back(i);
  if (!diff.empty() || got[0].nseg != want.nseg) {
    // the tolerated class, or a failure
    const std::string what = "nseg " + std::to_string(got[0].nseg) + " vs " + std::to_string(want.nseg) + ", " +
                             std::to_string(diff.size()) + " rows of iwhere differ (first: " +
                             (diff.empty() ? std::string("-") : std::to_string(diff[0])) + ")";
    if (diff.empty() || (long)diff.size() != std::labs((long)got[0].nseg - (long)want.nseg)) return what;
    double tb0 = -1.0;
    for (int i : diff) {
      // (its breakpoint from the row's own data, with the iwhere the scan leaves: 0)
      const double tb = lbk::brk_time_h<double>(c.x[i], c.l[i], c.u[i], c.nbd[i], c.g[i], 0);
      if (!(tb > 0.0) || tb == lbk::INF) return what + " -- a differing row has no breakpoint";
      if (tb0 < 0.0) tb0 = tb;
      if (std::fabs(tb - tb0) > 1e-12 * tb0) return what + " -- differing rows at different breakpoints";
      const int a = iw[i], b = want.iwhere[i];
      if (!((a == 0 && (b == 1 || b == 2)) || (b == 0 && (a == 1 || a == 2)))) return what + " -- not a fixed / free flip";
      // where the walk ended: a free row's xcp = x + tsum d, so tsum = (xcp - x) / d on the side that left it free
      const double xc = a == 0 ? xcp[i] : want.xcp[i];
      const double tsum = (xc - c.x[i]) / (-c.g[i]);
      if (std::fabs(tsum - tb0) > 1e-9 * tb0) return what + " -- the walk did not end on that breakpoint";
    }
    return "~" + what;
  }
  for (int r = 0; r < op.nranks; ++r)
    for (int k = 0; k < 4; ++k)
      if (!close_to(&got[r].pcwv[(size_t)2 * m * k], &want.pcwv[(size_t)2 * m * k], (size_t)2 * c.col, 1e-9, worst))
        return std::string("work vector ") + "pcwv"[k] + " off by " + std::to_string(worst) + " on rank " + std::to_string(r);
  return "";
}

// ============================================================ task 'START' over several ranks
// The bound dictionary is built by probing passes whose results are reduced over the ranks: every rank must end
// with the SAME tables after the same number of collectives whatever its own rows hold (a rank may see one value
// only, or none of the values another rank sees), the codes must decode to the row's own l, u, nbd, and a ninth
// value anywhere must make every rank fall back.  Also: the uniform flags, active's projection and counts.
static std::string check_start(std::mt19937_64 &rng, int n, int nranks, int kl, int ku, bool nbd_uniform) {
  using S = Solver<double>;
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<double> lv(kl), uv(ku), l(n), u(n), x(n);
  for (int j = 0; j < kl; ++j) lv[j] = -1.0 - 0.25 * j;
  for (int j = 0; j < ku; ++j) uv[j] = 1.0 + 0.5 * j;
  std::vector<int32_t> nbd(n);
  for (int i = 0; i < n; ++i) {
    // (blocks of rows share values, so that ranks see different subsets)
    const int blk = (int)((int64_t)i * 5 / n);
    l[i] = lv[(blk + (int)(U(rng) * 2)) % kl], u[i] = uv[(blk * 3 + (int)(U(rng) * 2)) % ku];
    nbd[i] = nbd_uniform ? 2 : (int)(U(rng) * 4.0) & 3;
    x[i] = 4.0 * (U(rng) - 0.5);
  }
  for (int j = 0; j < kl; ++j) l[(size_t)(rng() % n)] = lv[j];
  for (int j = 0; j < ku; ++j) u[(size_t)(rng() % n)] = uv[j];
  std::set<double> sl(l.begin(), l.end()), su(u.begin(), u.end());
  const bool want_dict = sl.size() <= 8 && su.size() <= 8 && !(sl.size() == 1 && su.size() == 1);
  World w;
  w.nranks = nranks, w.bar.n = nranks, w.src.assign(nranks, nullptr);
  std::vector<std::string> errs(nranks);
  std::vector<int> masks(nranks, -1);
  std::vector<lbk::BoundTables> tabs(nranks);
  std::vector<int64_t> colls(nranks, 0);
  auto body = [&](int rank) {
    const int64_t lo = (int64_t)n * rank / nranks, hi = (int64_t)n * (rank + 1) / nranks, nl = hi - lo;
    S *s = new S();
    int rc = s->init(nl, n, lo, 3, 0, 0, nullptr);
    RankComm rcomm{&w, rank};
    if (!rc && nranks > 1) rc = s->attach_host(cb_allreduce, cb_allgather, &rcomm, rank, nranks);
    if (!rc) rc = s->set_option("spin", 0.0);
    // (every rank's rows in buffers of their own: the entry asks for 16-byte aligned vectors)
    std::vector<double> xr(x.begin() + lo, x.begin() + hi), lr(l.begin() + lo, l.begin() + hi),
        ur(u.begin() + lo, u.begin() + hi), g(nl, 0.0), dsave(29, 0.0);
    std::vector<int32_t> nbr(nbd.begin() + lo, nbd.begin() + hi), lsave(4, 0), isave(44, 0);
    char task[60], csave[60];
    std::memset(task, ' ', 60), std::memset(csave, ' ', 60);
    std::memcpy(task, "START", 5);
    double f = 0.0;
    if (!rc)
      rc = s->setulb_dev(xr.data(), lr.data(), ur.data(), nbr.data(), &f, g.data(), 0.0, 0.0, task, -1, csave,
                         lsave.data(), isave.data(), dsave.data());
    if (rc) {
      errs[rank] = g_err;
    } else if (std::strncmp(task, "FG_START", 8) != 0) {
      errs[rank] = std::string("task ") + std::string(task, 20);
    } else {
      masks[rank] = s->ub_mask, tabs[rank] = s->ub_tab, colls[rank] = s->ncoll;
      for (int64_t i = 0; i < nl && errs[rank].empty(); ++i) {
        const double xi = std::min(std::max(x[lo + i], nbd[lo + i] != 0 && nbd[lo + i] <= 2 ? l[lo + i] : -1e300),
                                   nbd[lo + i] >= 2 ? u[lo + i] : 1e300);
        if (xr[i] != xi) errs[rank] = "active: x not projected as the reference projects it";
        if (s->ub_mask & lbk::UB_DICT) {
          const unsigned c = (unsigned)(unsigned char)s->nbd8[i];
          if ((int)(c & 3u) != nbd[lo + i] || s->ub_tab.l[(c >> 2) & 7u] != l[lo + i] || s->ub_tab.u[c >> 5] != u[lo + i])
            errs[rank] = "code byte of row " + std::to_string(lo + i) + " does not decode to its l, u, nbd";
        } else if (!(s->ub_mask & 4) && (int)s->nbd8[i] != nbd[lo + i]) {
          errs[rank] = "packed nbd";
        }
      }
    }
    delete s;
  };
  if (nranks == 1) {
    body(0);
  } else {
    std::vector<std::thread> th;
    for (int r = 0; r < nranks; ++r) th.emplace_back(body, r);
    for (auto &t : th) t.join();
  }
  for (int r = 0; r < nranks; ++r)
    if (!errs[r].empty()) return "rank " + std::to_string(r) + ": " + errs[r];
  for (int r = 0; r < nranks; ++r) {
    if (((masks[r] & lbk::UB_DICT) != 0) != want_dict)
      return "rank " + std::to_string(r) + ": mask " + std::to_string(masks[r]) + ", dictionary expected: " + std::to_string(want_dict);
    if (colls[r] != colls[0]) return "the ranks issued different numbers of collectives";
    // (nb0, the value of a uniform nbd array, is each rank's own)
    if (want_dict && (tabs[r].nl != tabs[0].nl || tabs[r].nu != tabs[0].nu ||
                      std::memcmp(tabs[r].l, tabs[0].l, sizeof tabs[0].l) != 0 ||
                      std::memcmp(tabs[r].u, tabs[0].u, sizeof tabs[0].u) != 0))
      return "the ranks hold different tables";
  }
  if (want_dict && (tabs[0].nl != (int)sl.size() || tabs[0].nu != (int)su.size())) return "table sizes";
  return "";
}

// a random bounded problem, driven by the ORACLE; every NEW_X state (and the start) becomes a case
struct Problem {
  int n, m;
  std::vector<double> a, cc, l, u, x0;
  std::vector<int> nbd;
  bool wavy;
  double fg(const double *x, double *g) const {
    double f = 0.0;
    for (int i = 0; i < n; ++i) {
      const double d = x[i] - cc[i];
      f += 0.5 * a[i] * d * d;
      g[i] = a[i] * d;
      if (wavy) f += std::cos(3 * x[i]), g[i] -= 3 * std::sin(3 * x[i]);
    }
    return f;
  }
};
static Problem make_problem(std::mt19937_64 &rng, int n, int m, int family) {
  Problem p;
  p.n = n, p.m = m, p.wavy = family == 1;
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> N(0.0, 1.0);
  p.a.resize(n), p.cc.resize(n), p.l.resize(n), p.u.resize(n), p.x0.resize(n), p.nbd.resize(n);
  const bool lattice = family == 2;  // values on a coarse grid: many EQUAL breakpoints (ties, heap order)
  auto q8 = [&](double v) { return lattice ? std::round(v * 4.0) / 4.0 : v; };
  for (int i = 0; i < n; ++i) {
    p.a[i] = lattice ? (double)(1 + (int)(U(rng) * 3)) : 1.0 + 99.0 * U(rng);
    p.cc[i] = q8(2.0 * N(rng));
    p.l[i] = q8(-1.0 + N(rng));
    p.u[i] = p.l[i] + q8(std::fabs(1.5 + N(rng))) + (lattice ? 0.25 : 0.0);
    if (U(rng) < 0.03) p.u[i] = p.l[i];
    p.nbd[i] = (int)(U(rng) * 4.0) & 3;
    p.x0[i] = q8(3.0 * N(rng));
  }
  if (family == 3)  // a plain box, every variable bounded on both sides: walks that fix (almost) every variable
    for (int i = 0; i < n; ++i) p.nbd[i] = 2;
  return p;
}
struct Tally {
  long cases = 0, failed = 0, walks_long = 0, nseg_total = 0, on_breakpoint = 0, multi_rank = 0, exact_order = 0;
};
static void run_problem(std::mt19937_64 &rng, const Problem &p, int max_iter, const std::vector<Opts> &modes, Tally &tl,
                        int seed_tag, bool modes_in_order = false) {
  const int n = p.n, m = p.m;
  std::vector<double> x = p.x0, g(n, 0.0), wa((size_t)lbo_wa_len(n, m), 0.0), dsave(29, 0.0);
  std::vector<int> iwa((size_t)3 * n, 0), lsave(4, 0), isave(44, 0);
  char task[60], csave[60];
  std::memset(task, ' ', 60), std::memset(csave, ' ', 60);
  std::memcpy(task, "START", 5);
  double f = 0.0;
  int64_t off[13];
  lbo_wa_offsets(n, m, off);
  size_t mode_at = (size_t)rng();
  if (modes_in_order) mode_at = 0;
  for (int calls = 0; calls < 100000; ++calls) {
    lbo_setulb(n, m, x.data(), p.l.data(), p.u.data(), p.nbd.data(), &f, g.data(), 0.0, 0.0, wa.data(), iwa.data(), task, -1,
               csave, lsave.data(), isave.data(), dsave.data());
    if (std::strncmp(task, "FG", 2) == 0) {
      f = p.fg(x.data(), g.data());
      if (std::strncmp(task, "FG_ST", 5) != 0) continue;
    } else if (std::strncmp(task, "NEW_X", 5) != 0) {
      break;
    }
    // a state cauchy can start from: x, g, iwhere and the L-BFGS matrices as this return leaves them (the pair
    // of the step just taken is not in W yet: a consistent, one-iteration-old model -- any such model will do)
    Case c;
    c.n = n, c.m = m, c.col = isave[27], c.head = isave[26], c.theta = dsave[0];
    const bool start = std::strncmp(task, "FG_ST", 5) == 0;
    if (start) c.col = 0, c.head = 1, c.theta = 1.0;
    c.x = x, c.l = p.l, c.u = p.u, c.g = g, c.nbd = p.nbd;
    c.ws.assign(wa.begin() + off[0], wa.begin() + off[0] + (size_t)m * n);
    c.wy.assign(wa.begin() + off[1], wa.begin() + off[1] + (size_t)m * n);
    c.sy.assign(wa.begin() + off[2], wa.begin() + off[2] + (size_t)m * m);
    c.wt.assign(wa.begin() + off[4], wa.begin() + off[4] + (size_t)m * m);
    c.iwhere.assign(iwa.begin() + n, iwa.begin() + 2 * n);
    c.sbgnrm = start ? (double)lbo_projgr(n, p.l.data(), p.u.data(), p.nbd.data(), x.data(), g.data()) : dsave[12];
    const Opts &op = modes[mode_at++ % modes.size()];
    const std::string why = check_case(c, op);
    Out o;
    run_oracle(c, o);
    tl.cases++, tl.nseg_total += o.nseg, tl.walks_long += o.nseg > 256;
    tl.multi_rank += op.nranks > 1, tl.exact_order += op.exact_always;
    if (!why.empty() && why[0] == '~') {
      tl.on_breakpoint++;
      if (std::getenv("WALK_CHECK_VERBOSE"))
        std::fprintf(stderr, "stationary point on a breakpoint: problem %d (n %d m %d) col %d ranks %d: %s\n", seed_tag, n, m,
                     c.col, op.nranks, why.c_str() + 1);
    } else if (!why.empty()) {
      tl.failed++;
      std::fprintf(stderr, "FAIL problem %d (n %d m %d) state col %d: ranks %d comm %d exact %d door %d base %lld: %s\n",
                   seed_tag, n, m, c.col, op.nranks, op.comm_kind, (int)op.exact_always, (int)op.via_door,
                   (long long)op.row_base, why.c_str());
    }
    if (isave[29] >= max_iter) break;
  }
}

int main(int argc, char **argv) {
  const int nproblems = argc > 1 ? std::atoi(argv[1]) : 60;
  const int only = argc > 2 ? std::atoi(argv[2]) : -1;  // (debugging: run this problem alone)
  // the RCCL entry points of the library, for the communicator route
  g_rccl.AllGather = fake_allgather, g_rccl.CommDestroy = fake_destroy, g_rccl.ok = true;
  std::vector<Opts> modes;
  {
    Opts o;
    modes.push_back(o);  // one rank, through import_state / the door / export_state
    o.exact_always = true;
    modes.push_back(o);  // ... every walk in the reference's heap order (exact_init / refill_exact)
    o = Opts{};
    o.via_door = false, o.row_base = ((int64_t)1 << 31) - 7;
    modes.push_back(o);  // row numbers across 2^31
    o.row_base = ((int64_t)1 << 32) - 5, o.exact_always = true;
    modes.push_back(o);  // ... across 2^32: the heap carries 64-bit row numbers
    for (int nr : {2, 3, 5})
      for (int ck : {0, 1}) {
        o = Opts{};
        o.nranks = nr, o.comm_kind = ck;
        modes.push_back(o);  // ragged row blocks; host merge (callbacks) / device merge (communicator)
        o.exact_always = true;
        modes.push_back(o);  // ... the replicated heap, records gathered by their owners
      }
    o = Opts{};
    o.nranks = 4, o.comm_kind = 1, o.row_base = ((int64_t)1 << 32) - 1000;
    modes.push_back(o);
  }
  Tally tl;
  std::mt19937_64 rng(20251005);
  for (int k = 0; k < nproblems; ++k) {
    // sizes: mostly small (many states, every mode), some with first walks of thousands of segments (windows of
    // more than 256 candidates, sorted lists, refills at 64 / 256 / 1024 / 4096 records), one beyond 2^18
    // candidates (the full sort of all breakpoints, the cursor form of the rows a walk fixes)
    int n = 40 + (int)(rng() % 400), m = 1 + (int)(rng() % 12), iters = 8;
    if (k % 5 == 3) n = 3000 + (int)(rng() % 9000), iters = 3;
    if (k % 20 == 11) n = 70000 + (int)(rng() % 20000), m = 4, iters = 2;
    if (k == 17) n = 480000, m = 3, iters = 1;
    if (k % 7 == 5) m = 13 + (int)(rng() % 19);  // (up to 31 pairs: MC = 20 / 32 result layouts of the scan)
    const Problem p = make_problem(rng, n, m, k == 17 ? 3 : k % 4);  // (17: a plain box, > 2^18 breakpoints)
    if (only >= 0 && k != only) {
      (void)rng();  // (keep the stream of the other problems' mode choices out of step-dependence)
      continue;
    }
    if (k == 17) {
      // the big one, in fixed modes: one rank (> 2^18 candidates in the first window: the full sort of all
      // breakpoints, chunks of up to 16 384 records with the next one prefetched, the register-resident loop, the
      // cursor form of the rows a walk fixes), then three ranks through the communicator route
      Opts a, b;
      b.nranks = 3, b.comm_kind = 1, b.via_door = false;
      run_problem(rng, p, iters, std::vector<Opts>{a, b}, tl, k, true);
      continue;
    }
    run_problem(rng, p, iters, modes, tl, k);
  }
  // ---- START over 1 - 5 ranks: uniform, few-valued (2 ... 8 values, one side possibly uniform) and 9-valued arrays ----
  long starts = 0, starts_failed = 0;
  if (only < 0)
    for (int rep = 0; rep < 60; ++rep) {
      const int nr = 1 + rep % 5, kl = 1 + (int)(rng() % 9), ku = 1 + (int)(rng() % 9);
      if (std::getenv("WALK_CHECK_VERBOSE")) std::fprintf(stderr, "start %d: ranks %d, %d / %d values\n", rep, nr, kl, ku);
      const std::string why = check_start(rng, 200 + (int)(rng() % 3000), nr, kl, ku, rep % 3 == 0);
      starts++;
      if (!why.empty()) {
        starts_failed++;
        std::fprintf(stderr, "FAIL start %d (ranks %d, %d / %d values): %s\n", rep, nr, kl, ku, why.c_str());
      }
    }
  tl.failed += starts_failed;
  std::printf("walk_check: %ld START cases over 1-5 ranks, %ld failed\n", starts, starts_failed);
  std::printf("walk_check: %ld cases, %ld failed, %ld stationary-point-on-a-breakpoint, %ld multi-rank, %ld in heap order, "
              "%ld walks of more than 256 segments, %ld segments in all; rank 0: %ld full sorts, %ld tie splits replayed, "
              "%ld host syncs, %ld collectives\n", tl.cases, tl.failed, tl.on_breakpoint, tl.multi_rank, tl.exact_order,
              tl.walks_long, tl.nseg_total, g_fullsorts.load(), g_tiesplits.load(), g_syncs.load(), g_collectives.load());
  return tl.failed == 0 && tl.cases > 0 ? 0 : 1;
}
