// k_misc.hip -- launch sizing, finalize, active/errclb, projgr, W'v, line-search vectors, objectives
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

#include <cstdio>
#include <mutex>
#include <unordered_map>

namespace lbk {

int grid_for(int64_t n, int vec) {
  int64_t g = (n / vec + BLOCK - 1) / BLOCK;
  if (g < 1) g = 1;
  if (g > MAX_BLOCKS) g = MAX_BLOCKS;
  return (int)g;
}

// The passes over W run on a grid of exactly the workgroups that are RESIDENT -- every workgroup strides
// over its share of the rows, none waits for a slot: 3 per CU (768) for the kernels that hold <= 170
// registers, 1 per CU (256) for the update pass with its 95 accumulators (334 registers: one wave per
// SIMD).  Round 3 launched 768 everywhere (sweep at n = 1e8 with the old reduction epilogue: 512 and 768
// equal, 1024+ slower); for a one-wave-per-SIMD kernel that is three ROUNDS of workgroups, each with its
// own ramp-up and tail during which its SIMD has no load in flight: update_scan at 1.25e7 rows 0.365 ->
// 0.329 ms with 256 (profiles/r4f).  The occupancy comes from the runtime, per kernel, once.
// Tune::wgrid > 0 overrides (at most MAX_BLOCKS - 1: the pair-shared update pass puts its leftover rows
// into one more column of the partial sums).
int grid_for_w(const Queue &q, int64_t n, int vec, const void *kernel) {
  const int g = grid_for(n, vec);
  int cap = q.tune.wgrid;
  if (cap <= 0) {
    static std::mutex mu;
    static std::unordered_map<const void *, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(kernel);
    if (it == cache.end()) {
      int per_cu = 0, cus = 0, dev = 0;
      if (!kernel || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, 0) != hipSuccess)
        per_cu = 3, (void)hipGetLastError();
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256, (void)hipGetLastError();
      per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
      if (std::getenv("LBFGSB_DEBUG")) {
        hipFuncAttributes fa{};
        if (kernel && hipFuncGetAttributes(&fa, kernel) != hipSuccess) (void)hipGetLastError();
        std::fprintf(stderr, "[grid] kernel %p: %d workgroups per CU x %d CUs (numRegs %d)\n", kernel, per_cu, cus,
                     fa.numRegs);
      }
      it = cache.emplace(kernel, per_cu * cus).first;
    }
    cap = it->second;
  }
  if (cap < 1) cap = 1;
  if (cap > MAX_BLOCKS - 1) cap = MAX_BLOCKS - 1;
  return g > cap ? cap : g;
}

bool pipe_on(const Queue &q, int mc, int elem_bytes) {
  if (q.tune.pipe == 0) return false;
  return mc == 20 || (mc == 10 && elem_bytes == 4);  // (the shapes DISPATCH_PIPE compiles)
}

int maxc_for(int col) { return col <= 5 ? 5 : (col <= 10 ? 10 : (col <= 20 ? 20 : 32)); }

// =========================== finalize ======================================
// One workgroup per output slot: fixed-order sum / min / max of the per-block partials.
// One launch serves up to three JOBS (partial-sum matrices of different kernels): a kernel whose results
// nobody waits for yet -- the caller's objective whose f is fetched with the next call, the storing pass
// of a deferred line-search set-up -- leaves its partials in a matrix of its own (Queue::part_sel) and
// PARKS its job (Queue::hold_fin); the next finalize launch takes the parked jobs along.  Every kernel
// boundary costs ~10 us on this machine (4 us for the tiny kernel + ~6 us of dependent-dispatch latency,
// profiles/r4k_iteration_timelines.txt): an iteration drops from 7 launches to 4.
// hpub != nullptr: the launch also PUBLISHES -- the last workgroup to finish copies the launch's results into
// mapped host memory (same offsets as in `res`) and then stores the launch's sequence number into a host
// word the host polls (fetch, solver.hip): no separate publish kernel for a single-rank context.
struct FinJobs {
  FinJob j[3];
  int n;
};
__global__ __launch_bounds__(BLOCK) void finalize_kernel(FinJobs J, double *res, double *hpub,
                                                         unsigned long long seq, unsigned long long *flag,
                                                         unsigned int *counter) {
  __shared__ double sm[BLOCK];
  int k = blockIdx.x, ji = 0;
  while (ji < J.n - 1 && k >= J.j[ji].nsum + J.j[ji].nmin + J.j[ji].nmax) {
    k -= J.j[ji].nsum + J.j[ji].nmin + J.j[ji].nmax;
    ++ji;
  }
  const double *__restrict__ part = J.j[ji].part;
  const int pstride = J.j[ji].pstride, nblocks = J.j[ji].nblocks, nsum = J.j[ji].nsum, nmin = J.j[ji].nmin;
  const int op = k < nsum ? 0 : (k < nsum + nmin ? 1 : 2);
  const double ident = op == 0 ? 0.0 : (op == 1 ? LB_INF : -LB_INF);
  double v = ident;
  if (nblocks <= 8 * BLOCK) {
    for (int b = threadIdx.x; b < nblocks; b += BLOCK) {
      const double p = part[(size_t)k * pstride + b];
      v = op == 0 ? v + p : (op == 1 ? fmin(v, p) : fmax(v, p));
    }
  } else {
    // very many partials (a pass launched with one trip per workgroup): eight independent chains per
    // thread, so that eight loads are in flight instead of one (fixed association all the same)
    double c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = ident;
    int b = threadIdx.x;
    for (; b + 7 * BLOCK < nblocks; b += 8 * BLOCK) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const double p = part[(size_t)k * pstride + b + e * BLOCK];
        c[e] = op == 0 ? c[e] + p : (op == 1 ? fmin(c[e], p) : fmax(c[e], p));
      }
    }
    for (int e = 0; b < nblocks; b += BLOCK, ++e) {
      const double p = part[(size_t)k * pstride + b];
      c[e & 7] = op == 0 ? c[e & 7] + p : (op == 1 ? fmin(c[e & 7], p) : fmax(c[e & 7], p));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v = op == 0 ? v + c[e] : (op == 1 ? fmin(v, c[e]) : fmax(v, c[e]));
  }
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double a = sm[threadIdx.x], b = sm[threadIdx.x + s];
      sm[threadIdx.x] = op == 0 ? a + b : (op == 1 ? fmin(a, b) : fmax(a, b));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) res[J.j[ji].off + k] = sm[0];
  if (hpub) {
    // The host must never see the sequence word before a result.  Results written to host memory by
    // workgroups on different XCDs travel different ways to the PCIe port, so the mirror is NOT written
    // slot by slot by whoever computed it: the LAST workgroup to finish (device-scope count) copies every
    // slot of the launch from d_res into the mirror -- one workgroup, one path -- fences at system scope and
    // then stores the word.
    __shared__ bool last;
    if (threadIdx.x == 0) {
      __threadfence();  // this workgroup's result is visible device-wide before it is counted
      const unsigned int done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      last = done == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
      __threadfence();
      for (int q = 0; q < J.n; ++q) {
        const int cnt = J.j[q].nsum + J.j[q].nmin + J.j[q].nmax, off = J.j[q].off;
        for (int e = threadIdx.x; e < cnt; e += BLOCK) {
          const unsigned long long bits = __hip_atomic_load(reinterpret_cast<unsigned long long *>(res + off + e),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          reinterpret_cast<unsigned long long *>(hpub)[off + e] = bits;
        }
      }
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

static void finalize_launch(Queue &q, const FinJobs &J) {
  int total = 0;
  for (int k = 0; k < J.n; ++k) total += J.j[k].nsum + J.j[k].nmin + J.j[k].nmax;
  if (total <= 0) return;
  const bool pub = q.fin_publish && q.hd_pub && q.hd_fin_flag && q.d_fin_count;
  if (pub) ++q.fin_seq;
  hipLaunchKernelGGL(finalize_kernel, dim3(total), dim3(BLOCK), 0, q.stream, J, q.d_res, pub ? q.hd_pub : nullptr,
                     q.fin_seq, q.hd_fin_flag, q.d_fin_count);
  LB_LAUNCHED(q);
  if (pub) q.launches_at_fin = q.launches;
}
void finalize_flush(Queue &q) {  // parked jobs that no later launch has taken along
  if (q.nheld == 0) return;
  FinJobs J{};
  for (int k = 0; k < q.nheld; ++k) J.j[J.n++] = q.held[k];
  q.nheld = 0;
  finalize_launch(q, J);
}
void finalize_from(Queue &q, const double *part, int pstride, int nblocks, int nsum,
                          int nmin, int nmax) {
  const int k = nsum + nmin + nmax;
  if (k <= 0) return;
  if (nblocks > pstride) {  // a partial-sum matrix has pstride columns per slot: never read beyond
    if (q.launch_err == hipSuccess) q.launch_err = hipErrorInvalidValue, q.launch_err_where = "finalize: nblocks > pstride";
    return;
  }
  const FinJob job{part, pstride, nblocks, q.res_off, nsum, nmin, nmax};
  if (q.hold_fin) {  // nobody waits for these yet: the next launch takes them along
    q.hold_fin = false;
    if (q.nheld == 2) finalize_flush(q);
    q.held[q.nheld++] = job;
    return;
  }
  FinJobs J{};
  for (int h = 0; h < q.nheld; ++h) J.j[J.n++] = q.held[h];
  q.nheld = 0;
  J.j[J.n++] = job;
  finalize_launch(q, J);
}
void launch_finalize(Queue &q, int nblocks, int nsum, int nmin, int nmax) {
  finalize_from(q, q.part(), MAX_BLOCKS, nblocks, nsum, nmin, nmax);
}

// =========================== publish ========================================
// End of a phase: the finalized results (one rank's, or every rank's after the all-gather) go
// straight into host memory the device can address, followed by a sequence word the host polls --
// instead of a D2H copy command + hipStreamSynchronize (15.6 us for the bare round trip on this
// pool against 6.9 us for a polled word, profiles/r03g_sync_latency.txt).  The data are written with
// system-scope visibility BEFORE the word: every lane's stores, a system-scope fence, the workgroup
// barrier, then lane 0's release store.  One workgroup: count <= a few hundred doubles.
__global__ __launch_bounds__(BLOCK) void publish_kernel(const double *__restrict__ src, double *dst_host,
                                                        int count, unsigned long long seq,
                                                        unsigned long long *flag_host) {
  for (int k = threadIdx.x; k < count; k += BLOCK) dst_host[k] = src[k];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish(Queue &q, const double *src, double *dst_host, int count, unsigned long long seq,
                    unsigned long long *flag_host) {
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(BLOCK), 0, q.stream, src, dst_host, count, seq, flag_host);
  LB_LAUNCHED(q);
}

// =========================== active / errclb ================================
template <typename T>
__global__ __launch_bounds__(BLOCK) void active_kernel(int64_t n, T *x, const T *l, const T *u,
                                                       const int32_t *nbd, iw_t *iwhere,
                                                       int8_t *wasfree, double *part) {
  double acc[4] = {0, 0, 0, 0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W];
    int nb[W], iw[W];
    ld<W>(x + i, xv);
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ldi<W>(nbd + i, nb);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (nb[k] > 0) {
        if (nb[k] <= 2 && xv[k] <= lv[k]) {
          if (xv[k] < lv[k]) {
            acc[0] += 1.0;
            xv[k] = lv[k];
          }
          acc[3] += 1.0;
        } else if (nb[k] >= 2 && xv[k] >= uv[k]) {
          if (xv[k] > uv[k]) {
            acc[0] += 1.0;
            xv[k] = uv[k];
          }
          acc[3] += 1.0;
        }
      }
      if (nb[k] != 2) acc[2] += 1.0;
      if (nb[k] == 0) {
        iw[k] = -1;
      } else {
        acc[1] += 1.0;
        iw[k] = (nb[k] == 2 && uv[k] - lv[k] <= 0.0) ? 3 : 0;
      }
      wasfree[i + k] = 1;
    }
    st<W>(x + i, xv);
    sti<W>(iwhere + i, iw);
  });
  block_reduce_store<4>(acc, 4, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_active(Queue &q, int64_t n, T *x, const T *l, const T *u, const int32_t *nbd,
                   iw_t *iwhere, int8_t *wasfree) {
  const int g = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(active_kernel<T>, dim3(g), dim3(BLOCK), 0, q.stream, n, x, l, u, nbd,
                     iwhere, wasfree, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, g, 4, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void errclb_kernel(int64_t n, int64_t row0, const T *l,
                                                       const T *u, const int32_t *nbd,
                                                       double *part) {
  double acc[5] = {0, 0, 0, 0, 0};
  const double l0 = (double)l[0], u0 = (double)u[0];
  const int nb0 = nbd[0];
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double lv[W], uv[W];
    int nb[W];
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ldi<W>(nbd + i, nb);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const double gi = (double)(row0 + i + k + 1);
      if (nb[k] < 0 || nb[k] > 3) acc[0] = fmax(acc[0], gi);
      if (nb[k] == 2 && lv[k] > uv[k]) acc[1] = fmax(acc[1], gi);
      // uniform bounds: is every entry the first one, bit for bit?  (-0.0 vs 0.0 and NaN count as different)
      if (__double_as_longlong(lv[k]) != __double_as_longlong(l0)) acc[2] = 1.0;
      if (__double_as_longlong(uv[k]) != __double_as_longlong(u0)) acc[3] = 1.0;
      if (nb[k] != nb0) acc[4] = 1.0;
    }
  });
  block_reduce_store<5>(acc, 0, 0, 5, part, MAX_BLOCKS);
}
template <typename T>
void launch_errclb(Queue &q, int64_t n, int64_t row0, const T *l, const T *u,
                   const int32_t *nbd) {
  const int g = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(errclb_kernel<T>, dim3(g), dim3(BLOCK), 0, q.stream, n, row0, l, u, nbd,
                     q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, g, 0, 0, 5);
}


template <typename T>
__global__ __launch_bounds__(BLOCK) void projgr_kernel(int64_t n, const T *x, const T *l,
                                                       const T *u, const int32_t *nbd,
                                                       const T *g, double *part) {
  double acc[1] = {0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W];
    int nb[W];
    ld<W>(x + i, xv);
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ld<W>(g + i, gv);
    ldi<W>(nbd + i, nb);
#pragma unroll
    for (int k = 0; k < W; ++k) acc[0] = fmax(acc[0], proj_g(xv[k], lv[k], uv[k], nb[k], gv[k]));
  });
  block_reduce_store<1>(acc, 0, 0, 1, part, MAX_BLOCKS);
}
template <typename T>
void launch_projgr(Queue &q, int64_t n, const T *x, const T *l, const T *u, const int32_t *nbd,
                   const T *g) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(projgr_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, x, l, u, nbd, g,
                     q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 0, 0, 1);
}

// =========================== level-1 doors ===================================
// The reference's n-length level-1 BLAS call sites (src/lbfgsb_blas_module.F90:37-277 as called from
// src/lbfgsb.f90:720-722 d = z - x, :812-822 y = g - r / s = stp d, :816 / :2196 / :2244 / :2335 the dots) are
// terms of the fused passes in an iteration; these three kernels exist for the doors of SURVEY.md 8(b)(4)
// (lbfgsb_hip_vec_sub / _vec_scale / _dot) and for callers that want the same primitives on the
// context's stream and -- the dot -- reduced over its ranks.
template <typename T>
__global__ __launch_bounds__(BLOCK) void vec_sub_kernel(int64_t n, const T *a, const T *b, T *out) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double av[W], bv[W];
    ld<W>(a + i, av);
    ld<W>(b + i, bv);
#pragma unroll
    for (int k = 0; k < W; ++k) av[k] = av[k] - bv[k];  // (REAL32: exact in fp64, rounded once by the store)
    st<W>(out + i, av);
  });
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void vec_scale_kernel(int64_t n, double alpha, T *v) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double vv[W];
    ld<W>(v + i, vv);
#pragma unroll
    for (int k = 0; k < W; ++k) vv[k] = alpha * vv[k];
    st<W>(v + i, vv);
  });
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void dot_kernel(int64_t n, const T *a, const T *b, double *part) {
  double acc[1] = {0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double av[W], bv[W];
    ld<W>(a + i, av);
    ld<W>(b + i, bv);
#pragma unroll
    for (int k = 0; k < W; ++k) acc[0] += av[k] * bv[k];
  });
  block_reduce_store<1>(acc, 1, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_vec_sub(Queue &q, int64_t n, const T *a, const T *b, T *out) {
  hipLaunchKernelGGL(vec_sub_kernel<T>, dim3(grid_for(n, VecOf<T>::V)), dim3(BLOCK), 0, q.stream, n, a, b, out);
  LB_LAUNCHED(q);
}
template <typename T>
void launch_vec_scale(Queue &q, int64_t n, double alpha, T *v) {
  hipLaunchKernelGGL(vec_scale_kernel<T>, dim3(grid_for(n, VecOf<T>::V)), dim3(BLOCK), 0, q.stream, n, alpha, v);
  LB_LAUNCHED(q);
}
template <typename T>
void launch_dot(Queue &q, int64_t n, const T *a, const T *b) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(dot_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, a, b, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 1, 0, 0);
}

// =========================== W'v ============================================
// The WS/WY correction-pair matvec: out[j] = sum_i Wy(i,j) v_i,
// out[col+j] = sum_i Ws(i,j) v_i.  Algorithmic bytes (2 col + 1) n s.
// Per lane and trip: 2*MC + 1 independent 16-byte loads in flight.
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void wtv_kernel(int64_t n, const T *__restrict__ ws,
                                                    const T *__restrict__ wy,
                                                    const T *__restrict__ zero, int64_t ldw, int m,
                                                    int head, int col, const T *__restrict__ v,
                                                    double *part) {
  double acc[2 * MC];
#pragma unroll
  for (int k = 0; k < 2 * MC; ++k) acc[k] = 0.0;
  for_rows<T, RowsPer<T, MC>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double vv[W], a[MC][W], b[MC][W];
    ldx<W, NT>(v + i, vv);
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, col, head, m, ldw) + i;
      ld_col<T, W, NT>(j < col, wy + off, zero, a[j]);
      ld_col<T, W, NT>(j < col, ws + off, zero, b[j]);
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += a[j][k] * vv[k];
        acc[MC + j] += b[j][k] * vv[k];
      }
    }
  });
  // slots [0..MC) = Wy' v, [MC..2MC) = Ws' v; entries >= col are discarded by the host
  block_reduce_store<2 * MC>(acc, 2 * MC, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_wtv_nofinalize(Queue &q, int64_t n, WStore<T> w, int head, int col, const T *v) {
  const int g = grid_for_w(q, n, VecOf<T>::V);
  DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((wtv_kernel<T, MC, NTV>), dim3(g), dim3(BLOCK), 0, q.stream, n,
                                        w.ws, w.wy, w.zero, w.ld, w.m, head, col, v, q.part()));
  LB_LAUNCHED(q);
}
template <typename T>
void launch_wtv(Queue &q, int64_t n, WStore<T> w, int head, int col, const T *v) {
  launch_wtv_nofinalize(q, n, w, head, col, v);
  launch_finalize(q, grid_for_w(q, n, VecOf<T>::V), 2 * maxc_for(col), 0, 0);
}


// Store a pending pair into its W slot without a subspace pass (subsm skipped, from-scratch
// formk, ...): Wy(:,slot) = T(g - r), Ws(:,slot) = T(stp*d).
template <typename T>
__global__ __launch_bounds__(BLOCK) void pair_commit_kernel(int64_t n, const T *__restrict__ g,
                                                            const T *__restrict__ r,
                                                            const T *__restrict__ d, double stp,
                                                            T *cwy, T *cws) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double gv[W], rv[W], dv[W];
    ld<W>(g + i, gv);
    ld<W>(r + i, rv);
    ld<W>(d + i, dv);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      rv[k] = pend_y<T>(gv[k], rv[k]);
      dv[k] = pend_s<T>(dv[k], stp);
    }
    st<W>(cwy + i, rv);
    st<W>(cws + i, dv);
  });
}
__global__ __launch_bounds__(BLOCK) void nbd_pack_kernel(int64_t n, const int32_t *__restrict__ nbd,
                                                         nb_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (nb_t)nbd[i];
}
void launch_nbd_pack(Queue &q, int64_t n, const int32_t *nbd, nb_t *out) {
  hipLaunchKernelGGL(nbd_pack_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, q.stream, n, nbd, out);
  LB_LAUNCHED(q);
}

// ---- dictionary-coded bounds (kernels_common.hpp, UB_DICT): build, pack, verify ----
// One probe pass: how many l_i / u_i are NOT in the tables yet (bit-for-bit membership), and the smallest such
// value of each array.  The host adds those two values to its tables and probes again until nothing is left
// or a table would exceed 8 entries; every quantity is reduced over the ranks, so all ranks build the SAME
// tables with the same number of passes.  res: sum [0] #l outside, [1] #u outside | min [2] l value, [3] u value
template <typename T>
__device__ __forceinline__ int dict_find(const BoundTables &tb, int which, double v) {
  const long long b = __double_as_longlong(v);
  const int cnt = which ? tb.nu : tb.nl;
  for (int j = 0; j < 8; ++j)
    if (j < cnt && __double_as_longlong(which ? tb.u[j] : tb.l[j]) == b) return j;
  return -1;
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void dict_probe_kernel(int64_t n, const T *__restrict__ l,
                                                           const T *__restrict__ u, BoundTables tb,
                                                           double *part) {
  double acc[4] = {0.0, 0.0, LB_INF, LB_INF};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double lv[W], uv[W];
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (dict_find<T>(tb, 0, lv[k]) < 0) acc[0] += 1.0, acc[2] = fmin(acc[2], lv[k]);
      if (dict_find<T>(tb, 1, uv[k]) < 0) acc[1] += 1.0, acc[3] = fmin(acc[3], uv[k]);
    }
  });
  block_reduce_store<4>(acc, 2, 2, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_dict_probe(Queue &q, int64_t n, const T *l, const T *u, const BoundTables &tb) {
  const int g = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(dict_probe_kernel<T>, dim3(g), dim3(BLOCK), 0, q.stream, n, l, u, tb, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, g, 2, 2, 0);
}
// the code byte of every row: nbd | l-index << 2 | u-index << 5 (a value the tables do not hold cannot occur:
// the probe has just found none; such a row would get index 0 and be caught by bounds_verify_kernel)
template <typename T>
__global__ __launch_bounds__(BLOCK) void nbd_pack_dict_kernel(int64_t n, const int32_t *__restrict__ nbd,
                                                              const T *__restrict__ l, const T *__restrict__ u,
                                                              BoundTables tb, nb_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int jl = dict_find<T>(tb, 0, (double)l[i]), ju = dict_find<T>(tb, 1, (double)u[i]);
    out[i] = (nb_t)(unsigned char)((unsigned)(nbd[i] & 3) | ((unsigned)(jl < 0 ? 0 : jl) << 2) |
                                   ((unsigned)(ju < 0 ? 0 : ju) << 5));
  }
}
template <typename T>
void launch_nbd_pack_dict(Queue &q, int64_t n, const int32_t *nbd, const T *l, const T *u, const BoundTables &tb,
                          nb_t *out) {
  hipLaunchKernelGGL(nbd_pack_dict_kernel<T>, dim3(grid_for(n, 1)), dim3(BLOCK), 0, q.stream, n, nbd, l, u, tb,
                     out);
  LB_LAUNCHED(q);
}
// The reference re-reads l, u, nbd on every call (src/lbfgsb.f90:1270-1330, 2594-2622, 2789-2816); the passes
// over W read the context's snapshot of them instead (the packed nbd byte, constants or table entries for
// uniform / few-valued bound arrays).  This pass compares the caller's arrays with that snapshot, bit for
// bit: a caller that edits bounds in place during a run is NOTICED (task 'ERROR: BOUNDS CHANGED DURING RUN')
// instead of silently iterated on with stale bounds.  ub bits as in the passes; tb: the values the passes
// use where an array is not streamed.  res: sum [0] = rows that differ
template <typename T>
__global__ __launch_bounds__(BLOCK) void bounds_verify_kernel(int64_t n, const T *__restrict__ l,
                                                              const T *__restrict__ u,
                                                              const int32_t *__restrict__ nbd,
                                                              const nb_t *__restrict__ code, int ub, BoundTables tb,
                                                              double *part) {
  double acc[1] = {0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double lv[W], uv[W];
    int nb[W], cd[W];
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ldi<W>(nbd + i, nb);
    if (ub & 4) {
#pragma unroll
      for (int k = 0; k < W; ++k) cd[k] = 0;
    } else {
      ldi<W>(code + i, cd);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const unsigned c = (unsigned)cd[k] & 0xffu;
      bool bad;
      if (ub & UB_DICT) {
        bad = nb[k] != (int)(c & 3u) ||
              __double_as_longlong(lv[k]) != __double_as_longlong(tb.l[(c >> 2) & 7u]) ||
              __double_as_longlong(uv[k]) != __double_as_longlong(tb.u[c >> 5]);
      } else {
        bad = (ub & 4) ? nb[k] != tb.nb0 : nb[k] != cd[k];
        if (ub & 1) bad = bad || __double_as_longlong(lv[k]) != __double_as_longlong(tb.l[0]);
        if (ub & 2) bad = bad || __double_as_longlong(uv[k]) != __double_as_longlong(tb.u[0]);
      }
      if (bad) acc[0] += 1.0;
    }
  });
  block_reduce_store<1>(acc, 1, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_bounds_verify(Queue &q, int64_t n, const T *l, const T *u, const int32_t *nbd, const nb_t *code,
                          int ub, const BoundTables &tb) {
  const int g = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(bounds_verify_kernel<T>, dim3(g), dim3(BLOCK), 0, q.stream, n, l, u, nbd, code, ub, tb,
                     q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, g, 1, 0, 0);
}
// bit-for-bit comparison of two copies of (l, u, nbd) -- the host-pointer form, whose device copies were made at
// START: the caller's arrays are uploaded again from time to time and compared.  res: sum [0] = rows that differ
template <typename T>
__global__ __launch_bounds__(BLOCK) void bounds_same_kernel(int64_t n, const T *__restrict__ l0,
                                                            const T *__restrict__ u0,
                                                            const int32_t *__restrict__ nb0,
                                                            const T *__restrict__ l1, const T *__restrict__ u1,
                                                            const int32_t *__restrict__ nb1, double *part) {
  double acc[1] = {0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double la[W], ua[W], lb[W], ub_[W];
    int na[W], nb[W];
    ld<W>(l0 + i, la), ld<W>(u0 + i, ua), ldi<W>(nb0 + i, na);
    ld<W>(l1 + i, lb), ld<W>(u1 + i, ub_), ldi<W>(nb1 + i, nb);
#pragma unroll
    for (int k = 0; k < W; ++k)
      if (__double_as_longlong(la[k]) != __double_as_longlong(lb[k]) ||
          __double_as_longlong(ua[k]) != __double_as_longlong(ub_[k]) || na[k] != nb[k])
        acc[0] += 1.0;
  });
  block_reduce_store<1>(acc, 1, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_bounds_same(Queue &q, int64_t n, const T *l0, const T *u0, const int32_t *nb0, const T *l1,
                        const T *u1, const int32_t *nb1) {
  const int g = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(bounds_same_kernel<T>, dim3(g), dim3(BLOCK), 0, q.stream, n, l0, u0, nb0, l1, u1, nb1,
                     q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, g, 1, 0, 0);
}
// xnew != nullptr: the next trial point x = stp d + t (lnsrlb_step_kernel's expression, :2262-2263, on d as it is
// stored) goes out in the same pass -- xnew may be x itself (each row is read before it is written)
template <typename T>
__global__ __launch_bounds__(BLOCK) void dz_materialise_kernel(int64_t n, const T *x, const T *__restrict__ t,
                                                               T *__restrict__ d, T *z, T *xnew, double stp) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], tv[W], dv[W];
    ld<W>(x + i, xv);
    ld<W>(t + i, tv);
#pragma unroll
    for (int k = 0; k < W; ++k) dv[k] = xv[k] - tv[k];  // exactly subsm_update_kernel's d = z - x
    st<W>(d + i, dv);
    if (z) st<W>(z + i, xv);
    if (xnew) {
      double o[W];
#pragma unroll
      for (int k = 0; k < W; ++k) o[k] = stp * (double)(T)dv[k] + tv[k];
      st<W>(xnew + i, o);
    }
  });
}
template <typename T>
void launch_dz_materialise(Queue &q, int64_t n, const T *x, const T *t, T *d, T *z, T *xnew, double stp) {
  hipLaunchKernelGGL(dz_materialise_kernel<T>, dim3(grid_for(n, VecOf<T>::V)), dim3(BLOCK), 0, q.stream,
                     n, x, t, d, z, xnew, stp);
  LB_LAUNCHED(q);
}
template <typename T>
void launch_pair_commit(Queue &q, int64_t n, const T *g, const T *r, const T *d, Pend pe,
                        WStore<T> w, int head, int col) {
  const int gr = grid_for(n, VecOf<T>::V);
  const int64_t slot = (int64_t)((head - 1 + col - 1) % w.m) * w.ld;
  hipLaunchKernelGGL(pair_commit_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, g, r, d, pe.stp,
                     w.wy + slot, w.ws + slot);
  LB_LAUNCHED(q);
}

// The Cauchy point as a vector, by the same per-row rule the fused passes use (xcp_row).
template <typename T>
__global__ __launch_bounds__(BLOCK) void xcp_fill_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ g, const T *__restrict__ l,
    const T *__restrict__ u, const iw_t *__restrict__ iwhere, double tsum, T *dst) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], gv[W], lv[W], uv[W], out[W];
    int iw[W];
    ld<W>(x + i, xv);
    ld<W>(g + i, gv);
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ldi<W>(iwhere + i, iw);
#pragma unroll
    for (int k = 0; k < W; ++k) out[k] = xcp_row<T>(xv[k], gv[k], iw[k], lv[k], uv[k], tsum);
    st<W>(dst + i, out);
  });
}
template <typename T>
void launch_xcp_fill(Queue &q, int64_t n, const T *x, const T *g, const T *l, const T *u,
                     const iw_t *iwhere, double tsum, T *dst) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(xcp_fill_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, x, g, l, u, iwhere,
                     tsum, dst);
  LB_LAUNCHED(q);
}

// backtracking ratio of one free variable (:2842-2857); 2.0 = no restriction
__device__ __forceinline__ double bt_ratio(double dk, double x, double l, double u, int nb) {
  double c = 2.0;
  if (nb != 0) {
    if (dk < 0.0 && nb <= 2) {
      const double t2 = l - x;
      c = t2 >= 0.0 ? 0.0 : t2 / dk;
    } else if (dk > 0.0 && nb >= 2) {
      const double t2 = u - x;
      c = t2 <= 0.0 ? 0.0 : t2 / dk;
    }
  }
  return c;
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void subsm_alpha_kernel(int64_t n, int64_t row0,
                                                            const T *__restrict__ xp,
                                                            const T *__restrict__ r,
                                                            const T *__restrict__ l,
                                                            const T *__restrict__ u,
                                                            const int32_t *__restrict__ nbd,
                                                            const iw_t *__restrict__ iwhere,
                                                            int pass, double alpha, double *part) {
  double acc[1] = {pass == 0 ? 1.0 : LB_INF};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (iwhere[i] > 0) continue;
    const double c = bt_ratio((double)r[i], (double)xp[i], (double)l[i], (double)u[i], nbd[i]);
    if (pass == 0)
      acc[0] = fmin(acc[0], c);
    else if (c == alpha)
      acc[0] = fmin(acc[0], (double)(row0 + i));
  }
  block_reduce_store<1>(acc, 0, 1, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_subsm_alpha(Queue &q, int64_t n, const T *xp, const T *r, const T *l, const T *u,
                        const int32_t *nbd, const iw_t *iwhere) {
  const int gr = grid_for(n, 1);
  hipLaunchKernelGGL(subsm_alpha_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, (int64_t)0, xp,
                     r, l, u, nbd, iwhere, 0, 0.0, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 0, 1, 0);
}
template <typename T>
void launch_subsm_argalpha(Queue &q, int64_t n, int64_t row0, const T *xp, const T *r, const T *l,
                           const T *u, const int32_t *nbd, const iw_t *iwhere, double alpha) {
  const int gr = grid_for(n, 1);
  hipLaunchKernelGGL(subsm_alpha_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, xp, r, l,
                     u, nbd, iwhere, 1, alpha, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 0, 1, 0);
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void subsm_backtrack_kernel(int64_t n, int64_t row0, T *z,
                                                                const T *__restrict__ xp, T *r,
                                                                const T *__restrict__ l,
                                                                const T *__restrict__ u,
                                                                const iw_t *__restrict__ iwhere,
                                                                double alpha, int64_t ibd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double xk = (double)xp[i];
    if (iwhere[i] <= 0) {
      double dk = (double)r[i];
      if (alpha < 1.0 && row0 + i == ibd) {  // :2865-2875
        if (dk > 0.0) {
          xk = (double)u[i];
          dk = 0.0;
        } else if (dk < 0.0) {
          xk = (double)l[i];
          dk = 0.0;
        }
        r[i] = (T)dk;
      }
      xk = xk + alpha * dk;
    }
    z[i] = (T)xk;
  }
}
template <typename T>
void launch_subsm_backtrack(Queue &q, int64_t n, int64_t row0, T *z, const T *xp, T *r, const T *l,
                            const T *u, const iw_t *iwhere, double alpha, int64_t ibd) {
  const int gr = grid_for(n, 1);
  hipLaunchKernelGGL(subsm_backtrack_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, z, xp,
                     r, l, u, iwhere, alpha, ibd);
  LB_LAUNCHED(q);
}

// =========================== lnsrlb (:2174-2275) =============================
template <typename T>
__global__ __launch_bounds__(BLOCK) void lnsrlb_begin_kernel(
    int64_t n, const T *__restrict__ z, const T *__restrict__ x, const T *__restrict__ g,
    const T *__restrict__ l, const T *__restrict__ u, const int32_t *__restrict__ nbd, T *d, T *t,
    T *r, int do_stpmx, double *part) {
  double acc[3] = {0.0, 0.0, 1.0e10};  // dtd, gd, stpmx (big, :2189)
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double zv[W], xv[W], gv[W], dv[W], lv[W], uv[W];
    int nb[W];
    ld<W>(z + i, zv);
    ld<W>(x + i, xv);
    ld<W>(g + i, gv);
    if (do_stpmx) {
      ld<W>(l + i, lv);
      ld<W>(u + i, uv);
      ldi<W>(nbd + i, nb);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      dv[k] = zv[k] - xv[k];  // mainlb :720-722
      acc[0] = acc[0] + dv[k] * dv[k];
      acc[1] = acc[1] + gv[k] * dv[k];
      if (do_stpmx && nb[k] != 0) {  // :2206-2225
        const double a1 = dv[k];
        if (a1 < 0.0 && nb[k] <= 2) {
          const double a2 = lv[k] - xv[k];
          acc[2] = fmin(acc[2], a2 >= 0.0 ? 0.0 : a2 / a1);
        } else if (a1 > 0.0 && nb[k] >= 2) {
          const double a2 = uv[k] - xv[k];
          acc[2] = fmin(acc[2], a2 <= 0.0 ? 0.0 : a2 / a1);
        }
      }
    }
    st<W>(d + i, dv);
    // t, r == nullptr (ping-pong iterate buffers): x and g stay where they are and BECOME t and r
    if (t) st<W>(t + i, xv);
    if (r) st<W>(r + i, gv);
  });
  block_reduce_store<3>(acc, 2, 1, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_lnsrlb_begin(Queue &q, int64_t n, const T *z, const T *x, const T *g, const T *l,
                         const T *u, const int32_t *nbd, T *d, T *t, T *r, int do_stpmx) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(lnsrlb_begin_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, z, x, g, l, u,
                     nbd, d, t, r, do_stpmx, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 2, 1, 0);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void lnsrlb_step_kernel(int64_t n, T *x,
                                                            const T *__restrict__ z,
                                                            const T *__restrict__ d,
                                                            const T *__restrict__ t, double stp) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double o[W];
    if (stp == 1.0) {
      ld<W>(z + i, o);  // bit copy of z keeps exact bound values (:2264-2265)
    } else {
      double dv[W], tv[W];
      ld<W>(d + i, dv);
      ld<W>(t + i, tv);
#pragma unroll
      for (int k = 0; k < W; ++k) o[k] = stp * dv[k] + tv[k];
    }
    st<W>(x + i, o);
  });
}
template <typename T>
void launch_lnsrlb_step(Queue &q, int64_t n, T *x, const T *z, const T *d, const T *t,
                        double stp) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(lnsrlb_step_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, x, z, d, t, stp);
  LB_LAUNCHED(q);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void lnsrlb_eval_kernel(int64_t n, const T *__restrict__ x,
                                                            const T *__restrict__ l,
                                                            const T *__restrict__ u,
                                                            const int32_t *__restrict__ nbd,
                                                            const T *__restrict__ g,
                                                            const T *__restrict__ d, double *part) {
  double acc[2] = {0.0, 0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], dv[W];
    int nb[W];
    ld<W>(x + i, xv);
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ld<W>(g + i, gv);
    ld<W>(d + i, dv);
    ldi<W>(nbd + i, nb);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      acc[0] = acc[0] + gv[k] * dv[k];
      acc[1] = fmax(acc[1], proj_g(xv[k], lv[k], uv[k], nb[k], gv[k]));
    }
  });
  block_reduce_store<2>(acc, 1, 0, 1, part, MAX_BLOCKS);
}
template <typename T>
void launch_lnsrlb_eval(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                        const int32_t *nbd, const T *g, const T *d) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(lnsrlb_eval_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, x, l, u, nbd, g,
                     d, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 1, 0, 1);
}

// =========================== built-in objectives =============================
template <typename T>
__global__ __launch_bounds__(BLOCK) void obj_quadratic_kernel(int64_t n, int64_t row0,
                                                              const T *__restrict__ x, T *g,
                                                              int nt, double *part) {
  double acc[1] = {0.0};
  // The two residues of the problem's definition in 32-bit arithmetic where the row numbers allow it (the same
  // integers: (k gi) mod p = ((k mod p)(gi mod p)) mod p, 104729 = 4726 mod 100003, every product below 2^32):
  // two 64-bit remainders per row made this 16 B/row kernel instruction-bound (0.33 ms at n = 1e8 where its
  // bytes take 0.26).
  const bool fits32 = (uint64_t)(row0 + n + 1) < 0xffffffffull;  // (uniform)
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], gv[W];
    ld<W>(x + i, xv);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const int64_t gi = row0 + i + k + 1;
      uint32_t ra, rc;
      if (fits32) {
        const uint32_t g32 = (uint32_t)gi;
        ra = (7919u * (g32 % 10007u)) % 10007u;
        rc = (4726u * (g32 % 100003u)) % 100003u;
      } else {
        ra = (uint32_t)((7919 * gi) % 10007);
        rc = (uint32_t)((104729 * gi) % 100003);
      }
      const double a = 1.0 + 99.0 * (double)ra / 10006.0;
      const double c = -2.0 + 4.0 * (double)rc / 100002.0;
      const double dx = xv[k] - c;
      gv[k] = a * dx;
      acc[0] = acc[0] + a * dx * dx;
    }
    // large problems: stream g out, so that no dirty lines linger in the cache hierarchy and
    // drain into the read-only pass that follows
    if (nt)
      stnt<W>(g + i, gv);
    else
      st<W>(g + i, gv);
  });
  block_reduce_store<1>(acc, 1, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_obj_quadratic(Queue &q, int64_t n, int64_t row0, const T *x, T *g) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(obj_quadratic_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, x, g, q.nt ? 1 : 0,
                     q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 1, 0, 0);
}
// a caller's objective value that lives on the device: one "partial" for the fixed-order finalize, so that it travels
// like the value of a built-in objective (slot 0 of the next fetch)
__global__ void scalar_partial_kernel(const double *__restrict__ v, double *part) { part[0] = v[0]; }
void launch_scalar_partial(Queue &q, const double *d_val) {
  hipLaunchKernelGGL(scalar_partial_kernel, dim3(1), dim3(1), 0, q.stream, d_val, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, 1, 1, 0, 0);
}
// rows [row0, row0+n) of the chain; xl / xr = the neighbours' boundary elements x(row0-1),
// x(row0+n) (1-element halo, exchanged by the caller; unused at the global ends)
template <typename T>
__global__ __launch_bounds__(BLOCK) void obj_rosenbrock_kernel(int64_t n, int64_t row0,
                                                               int64_t nglob,
                                                               const T *__restrict__ x, T *g,
                                                               double xl, double xr, int nt,
                                                               double *part) {
  double acc[1] = {0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double xi = (double)x[i];
    const int64_t gi_ = row0 + i;
    double gi;
    if (gi_ == 0) {
      const double xp1 = i + 1 < n ? (double)x[i + 1] : xr;
      const double t1 = xp1 - xi * xi;
      gi = 2.0 * (xi - 1.0) - 16.0 * xi * t1;
      acc[0] = acc[0] + 0.25 * ((xi - 1.0) * (xi - 1.0));
    } else {
      const double xm = i > 0 ? (double)x[i - 1] : xl;
      const double t2 = xi - xm * xm;
      acc[0] = acc[0] + t2 * t2;
      if (gi_ == nglob - 1) {
        gi = 8.0 * t2;
      } else {
        const double xp1 = i + 1 < n ? (double)x[i + 1] : xr;
        const double t1 = xp1 - xi * xi;
        gi = 8.0 * t2 - 16.0 * xi * t1;
      }
    }
    if (nt)
      __builtin_nontemporal_store((T)gi, g + i);  // (see obj_quadratic_kernel)
    else
      g[i] = (T)gi;
  }
  block_reduce_store<1>(acc, 1, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_obj_rosenbrock(Queue &q, int64_t n, int64_t row0, int64_t nglob, const T *x, T *g,
                           double xl, double xr) {
  const int gr = grid_for(n, 1);
  hipLaunchKernelGGL(obj_rosenbrock_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, nglob,
                     x, g, xl, xr, q.nt ? 1 : 0, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 1, 0, 0);
}
// first and last local element, as doubles, into out[0..1] (halo message)
template <typename T>
__global__ void halo_pack_kernel(int64_t n, const T *__restrict__ x, double *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = (double)x[0];
    out[1] = (double)x[n - 1];
  }
}
template <typename T>
void launch_halo_pack(Queue &q, int64_t n, const T *x, double *out) {
  hipLaunchKernelGGL(halo_pack_kernel<T>, dim3(1), dim3(64), 0, q.stream, n, x, out);
  LB_LAUNCHED(q);
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_active<T>(Queue &, int64_t, T *, const T *, const T *, const int32_t *, iw_t *, int8_t *); \
  template void launch_errclb<T>(Queue &, int64_t, int64_t, const T *, const T *, const int32_t *); \
  template void launch_dict_probe<T>(Queue &, int64_t, const T *, const T *, const BoundTables &); \
  template void launch_nbd_pack_dict<T>(Queue &, int64_t, const int32_t *, const T *, const T *, const BoundTables &, nb_t *); \
  template void launch_bounds_verify<T>(Queue &, int64_t, const T *, const T *, const int32_t *, const nb_t *, int, const BoundTables &); \
  template void launch_bounds_same<T>(Queue &, int64_t, const T *, const T *, const int32_t *, const T *, const T *, const int32_t *); \
  template void launch_projgr<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *); \
  template void launch_vec_sub<T>(Queue &, int64_t, const T *, const T *, T *); \
  template void launch_vec_scale<T>(Queue &, int64_t, double, T *); \
  template void launch_dot<T>(Queue &, int64_t, const T *, const T *); \
  template void launch_wtv<T>(Queue &, int64_t, WStore<T>, int, int, const T *); \
  template void launch_wtv_nofinalize<T>(Queue &, int64_t, WStore<T>, int, int, const T *); \
  template void launch_xcp_fill<T>(Queue &, int64_t, const T *, const T *, const T *, const T *, const iw_t *, double, T *); \
  template void launch_subsm_alpha<T>(Queue &, int64_t, const T *, const T *, const T *, const T *, const int32_t *, const iw_t *); \
  template void launch_subsm_argalpha<T>(Queue &, int64_t, int64_t, const T *, const T *, const T *, const T *, const int32_t *, const iw_t *, double); \
  template void launch_subsm_backtrack<T>(Queue &, int64_t, int64_t, T *, const T *, T *, const T *, const T *, const iw_t *, double, int64_t); \
  template void launch_lnsrlb_begin<T>(Queue &, int64_t, const T *, const T *, const T *, const T *, const T *, const int32_t *, T *, T *, T *, int); \
  template void launch_lnsrlb_step<T>(Queue &, int64_t, T *, const T *, const T *, const T *, double); \
  template void launch_lnsrlb_eval<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, const T *); \
  template void launch_pair_commit<T>(Queue &, int64_t, const T *, const T *, const T *, Pend, WStore<T>, int, int); \
  template void launch_dz_materialise<T>(Queue &, int64_t, const T *, const T *, T *, T *, T *, double); \
  template void launch_obj_quadratic<T>(Queue &, int64_t, int64_t, const T *, T *); \
  template void launch_obj_rosenbrock<T>(Queue &, int64_t, int64_t, int64_t, const T *, T *, double, double); \
  template void launch_halo_pack<T>(Queue &, int64_t, const T *, double *);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
