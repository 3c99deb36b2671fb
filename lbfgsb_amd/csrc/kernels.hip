// kernels.hip -- hand-written gfx950 (CDNA4) kernels for the n-dimensional work
// of the L-BFGS-B iteration (reference src/lbfgsb.f90, routines cited per kernel).
//
// Shape of every kernel: tall-skinny, HBM-bound, no reuse.  One lane owns V
// consecutive rows (16 B per array per load, dwordx4), grid-stride over rows,
// <= 2048 workgroups of 4 wave64.  The correction-pair matrices Ws, Wy are
// column-major with a 256-byte aligned leading dimension, so lane i of a wave
// reads 16 B at column_base + 16*i: every wave-instruction is one fully
// coalesced 1 KiB request per column.  Reductions: per-lane fp64 accumulators
// -> wave shuffle -> LDS across the 4 waves -> one partial per workgroup ->
// fixed-order finalize kernel (deterministic; no float atomics).  2m <= 64
// columns is far too thin for MFMA: the flop/byte ratio is <= 2 (fp64), the
// machine balance ~10, so the roofline is HBM bandwidth.
//
// Column loops are unrolled to a compile-time MAXC; logical columns >= col are
// redirected to logical column 0 (an L1/L2 hit, no HBM traffic) and their
// results discarded, which keeps every load unconditional and in flight
// together.
#include "kernels.hpp"

#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "device_util.hpp"

namespace lbk {

#define LB_INF (__builtin_huge_val())

int grid_for(int64_t n, int vec) {
  // LBFGSB_GRID: cap on the number of workgroups (<= MAX_BLOCKS), for tuning experiments
  static const int cap = [] {
    const char *e = std::getenv("LBFGSB_GRID");
    const int v = e ? std::atoi(e) : 0;
    return v >= 1 && v <= MAX_BLOCKS ? v : MAX_BLOCKS;
  }();
  int64_t g = (n / vec + BLOCK - 1) / BLOCK;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// The passes over W keep 2-3 workgroups resident per CU (132-250 VGPRs): a grid of about that
// many workgroups, each striding over more rows, reads 3-6 % faster than 2048 of them (sweep at
// n = 1e8: 512 and 768 workgroups are equal, 1024+ slower).  LBFGSB_WGRID overrides.
// (fp32, m = 10 measured the other way round -- 2048: 150 it/s, 768: 144 -- and keeps 2048.)
int grid_for_w(int64_t n, int vec, int elem_bytes) {
  static const int env_cap = [] {
    const char *e = std::getenv("LBFGSB_WGRID");
    const int v = e ? std::atoi(e) : 0;
    return v >= 1 && v <= MAX_BLOCKS ? v : 0;
  }();
  const int cap = env_cap ? env_cap : (elem_bytes == 8 ? 768 : MAX_BLOCKS);
  const int g = grid_for(n, vec);
  return g > cap ? cap : g;
}

int maxc_for(int col) { return col <= 5 ? 5 : (col <= 10 ? 10 : (col <= 20 ? 20 : 32)); }

// the same, also selecting the load policy at run time (q.nt)
#define DISPATCH_MAXC_NT(col, ntflag, ...)   \
  do {                                        \
    if (ntflag) {                             \
      constexpr bool NTV = true;              \
      DISPATCH_MAXC(col, __VA_ARGS__);        \
    } else {                                  \
      constexpr bool NTV = false;             \
      DISPATCH_MAXC(col, __VA_ARGS__);        \
    }                                         \
  } while (0)

#define DISPATCH_MAXC(col, ...)       \
  do {                                \
    if ((col) <= 5) {                 \
      constexpr int MC = 5;           \
      __VA_ARGS__;                    \
    } else if ((col) <= 10) {         \
      constexpr int MC = 10;          \
      __VA_ARGS__;                    \
    } else if ((col) <= 20) {         \
      constexpr int MC = 20;          \
      __VA_ARGS__;                    \
    } else {                          \
      constexpr int MC = 32;          \
      __VA_ARGS__;                    \
    }                                 \
  } while (0)

// physical column offset (elements) of logical column j; j >= col -> logical 0
__device__ __forceinline__ int64_t col_off(int j, int col, int head, int m, int64_t ld) {
  const int jj = j < col ? j : 0;
  return (int64_t)((head - 1 + jj) % m) * ld;
}
// Unroll slots beyond the stored pairs (the kernels are unrolled to MC = 5/10/20/32 columns;
// e.g. the update pass at col - 1 = 9 old columns runs the MC = 10 code).  col_off sends such a
// slot to column 0 again, and with nontemporal loads that second request goes to HBM like the
// first (PMC: +1.3 GB per launch at n = 1e8).  The fp64 kernels therefore skip the load; the
// fp32 kernels, which already live at the register limit, lose 2x to the guarded form and keep
// the duplicate load.
template <typename T, int W, bool NT>
__device__ __forceinline__ void ld_col(bool live, const T *p, double (&o)[W]) {
  if constexpr (sizeof(T) == 8) {
    if (live) {
      ldx<W, NT>(p, o);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) o[k] = 0.0;
    }
  } else {
    ldx<W, NT>(p, o);
  }
}

// ---- pending pair ----
// Between matupd and the subspace pass of the same setulb call the newest pair (logical column
// col-1) is not in W yet: update_scan_kernel only reduces, so that it stays a read-only pass
// (a single store stream drops a streaming pass on MI355X from ~6.5 to ~4.8 TB/s,
// profiles/scripts/write_cost.hip).  Until subsm_update_kernel -- which stores vectors anyway --
// commits it, the column is defined by the vectors it was formed from, with the rounding of a
// store to T:   y = T(g - r),   s = T(stp * d)   (mainlb :813-822, matupd :2313-2314).
template <typename T>
__device__ __forceinline__ double pend_y(double gk, double rk) {
  return (double)(T)(gk - rk);
}
template <typename T>
__device__ __forceinline__ double pend_s(double dk, double stp) {
  return stp != 1.0 ? (double)(T)(stp * dk) : dk;
}
// columns j = 0..MC-1 of one row group; the pending column is read from (r, d) instead
template <typename T, int MC, int W, bool NT>
__device__ __forceinline__ void load_cols(const T *__restrict__ wy, const T *__restrict__ ws,
                                          const T *pr, const T *pd, int64_t i, int col, int head,
                                          int m, int64_t ldw, Pend pe, double (&a)[MC][W],
                                          double (&b)[MC][W]) {
  // one base pointer per matrix and a selected element offset (selecting between two base
  // pointers per column makes the compiler keep a table of addresses in scratch memory);
  // all buffers are allocations of T, so the distances are whole elements
  const int64_t dy = pe.on ? (int64_t)(((intptr_t)pr - (intptr_t)wy) / (intptr_t)sizeof(T)) : 0;
  const int64_t ds = pe.on ? (int64_t)(((intptr_t)pd - (intptr_t)ws) / (intptr_t)sizeof(T)) : 0;
#pragma unroll
  for (int j = 0; j < MC; ++j) {
    const int64_t off = col_off(j, col, head, m, ldw);
    const bool pj = pe.on && j == col - 1;
    ld_col<T, W, NT>(j < col, wy + ((pj ? dy : off) + i), a[j]);
    ld_col<T, W, NT>(j < col, ws + ((pj ? ds : off) + i), b[j]);
  }
}
template <typename T, int MC, int W>
__device__ __forceinline__ void fix_pending(int col, Pend pe, const double (&gv)[W],
                                            double (&a)[MC][W], double (&b)[MC][W]) {
  // branch-free selects: a predicated write a[col-1][k] = ... would turn the register arrays
  // into dynamically indexed ones (scratch memory)
#pragma unroll
  for (int j = 0; j < MC; ++j) {
    const bool pj = pe.on && j == col - 1;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const double yk = pend_y<T>(gv[k], a[j][k]);
      const double sk = pend_s<T>(b[j][k], pe.stp);
      a[j][k] = pj ? yk : a[j][k];
      b[j][k] = pj ? sk : b[j][k];
    }
  }
}

// =========================== finalize ======================================
// One workgroup per output slot: fixed-order sum / min / max of the per-block
// partials.
__global__ __launch_bounds__(BLOCK) void finalize_kernel(const double *__restrict__ part,
                                                         int pstride, int nblocks,
                                                         double *__restrict__ res, int nsum,
                                                         int nmin, int nmax) {
  __shared__ double sm[BLOCK];
  const int k = blockIdx.x;
  const int op = k < nsum ? 0 : (k < nsum + nmin ? 1 : 2);
  double v = op == 0 ? 0.0 : (op == 1 ? LB_INF : -LB_INF);
  for (int b = threadIdx.x; b < nblocks; b += BLOCK) {
    const double p = part[(size_t)k * pstride + b];
    v = op == 0 ? v + p : (op == 1 ? fmin(v, p) : fmax(v, p));
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
  if (threadIdx.x == 0) res[k] = sm[0];
}

static void finalize_from(Queue &q, const double *part, int pstride, int nblocks, int nsum,
                          int nmin, int nmax) {
  const int k = nsum + nmin + nmax;
  if (k <= 0) return;
  hipLaunchKernelGGL(finalize_kernel, dim3(k), dim3(BLOCK), 0, q.stream, part, pstride, nblocks,
                     q.d_res + q.res_off, nsum, nmin, nmax);
  q.launches++;
}
void launch_finalize(Queue &q, int nblocks, int nsum, int nmin, int nmax) {
  finalize_from(q, q.d_part, MAX_BLOCKS, nblocks, nsum, nmin, nmax);
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
                     iwhere, wasfree, q.d_part);
  q.launches++;
  launch_finalize(q, g, 4, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void errclb_kernel(int64_t n, int64_t row0, const T *l,
                                                       const T *u, const int32_t *nbd,
                                                       double *part) {
  double acc[2] = {0, 0};
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
    }
  });
  block_reduce_store<2>(acc, 0, 0, 2, part, MAX_BLOCKS);
}
template <typename T>
void launch_errclb(Queue &q, int64_t n, int64_t row0, const T *l, const T *u,
                   const int32_t *nbd) {
  const int g = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(errclb_kernel<T>, dim3(g), dim3(BLOCK), 0, q.stream, n, row0, l, u, nbd,
                     q.d_part);
  q.launches++;
  launch_finalize(q, g, 0, 0, 2);
}

// =========================== projgr (:2594-2622) ============================
__device__ __forceinline__ double proj_g(double x, double l, double u, int nb, double gi) {
  if (nb != 0) {
    if (gi < 0.0) {
      if (nb >= 2) gi = fmax(x - u, gi);
    } else {
      if (nb <= 2) gi = fmin(x - l, gi);
    }
  }
  return fabs(gi);
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
                     q.d_part);
  q.launches++;
  launch_finalize(q, gr, 0, 0, 1);
}

// =========================== W'v ============================================
// The WS/WY correction-pair matvec: out[j] = sum_i Wy(i,j) v_i,
// out[col+j] = sum_i Ws(i,j) v_i.  Algorithmic bytes (2 col + 1) n s.
// Per lane and trip: 2*MC + 1 independent 16-byte loads in flight.
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void wtv_kernel(int64_t n, const T *__restrict__ ws,
                                                    const T *__restrict__ wy, int64_t ldw, int m,
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
      ld_col<T, W, NT>(j < col, wy + off, a[j]);
      ld_col<T, W, NT>(j < col, ws + off, b[j]);
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
  const int g = grid_for_w(n, VecOf<T>::V, (int)sizeof(T));
  DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((wtv_kernel<T, MC, NTV>), dim3(g), dim3(BLOCK), 0, q.stream, n,
                                        w.ws, w.wy, w.ld, w.m, head, col, v, q.d_part));
  q.launches++;
}
template <typename T>
void launch_wtv(Queue &q, int64_t n, WStore<T> w, int head, int col, const T *v) {
  launch_wtv_nofinalize(q, n, w, head, col, v);
  launch_finalize(q, grid_for_w(n, VecOf<T>::V, (int)sizeof(T)), 2 * maxc_for(col), 0, 0);
}

// =========================== cauchy scan (:1270-1330) ========================
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void cauchy_scan_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const int32_t *__restrict__ nbd, const T *__restrict__ g, iw_t *iwhere, T *tbrk,
    const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw, int m, int head, int col,
    double *part) {
  constexpr int NA = 2 * MC + 5;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  acc[2 * MC + 4] = LB_INF;  // bkmin
  for_rows<T, RowsPer<T, MC>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], tb[W], ng[W];
    int nb[W], iw[W];
    ldx<W, NT>(x + i, xv);
    ldx<W, NT>(l + i, lv);
    ldx<W, NT>(u + i, uv);
    ldx<W, NT>(g + i, gv);
    ldi<W>(nbd + i, nb);
    ldi<W>(iwhere + i, iw);
    double a[MC > 0 ? MC : 1][W], b[MC > 0 ? MC : 1][W];
    if constexpr (MC > 0) {
#pragma unroll
      for (int j = 0; j < MC; ++j) {
        const int64_t off = col_off(j, col, head, m, ldw) + i;
        ld_col<T, W, NT>(j < col, wy + off, a[j]);
        ld_col<T, W, NT>(j < col, ws + off, b[j]);
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const double neggi = -gv[k];
      double tl = 0.0, tu = 0.0;
      if (iw[k] != 3 && iw[k] != -1) {
        if (nb[k] <= 2) tl = xv[k] - lv[k];
        if (nb[k] >= 2) tu = uv[k] - xv[k];
        const bool xlower = nb[k] <= 2 && tl <= 0.0;
        const bool xupper = nb[k] >= 2 && tu <= 0.0;
        iw[k] = 0;
        if (xlower) {
          if (neggi <= 0.0) iw[k] = 1;
        } else if (xupper) {
          if (neggi >= 0.0) iw[k] = 2;
        } else {
          if (fabs(neggi) <= 0.0) iw[k] = -3;
        }
      }
      if (iw[k] != 0 && iw[k] != -1) {
        tb[k] = -1.0;
        ng[k] = 0.0;
      } else {
        ng[k] = neggi;
        acc[2 * MC] = acc[2 * MC] - neggi * neggi;  // f1
        if (nb[k] <= 2 && nb[k] != 0 && neggi < 0.0) {
          tb[k] = tl / (-neggi);
          acc[2 * MC + 1] += 1.0;
          acc[2 * MC + 4] = fmin(acc[2 * MC + 4], tb[k]);
        } else if (nb[k] >= 2 && neggi > 0.0) {
          tb[k] = tu / neggi;
          acc[2 * MC + 1] += 1.0;
          acc[2 * MC + 4] = fmin(acc[2 * MC + 4], tb[k]);
        } else {
          tb[k] = LB_INF;
          acc[2 * MC + 2] += 1.0;
          if (fabs(neggi) > 0.0) acc[2 * MC + 3] += 1.0;
        }
      }
    }
    if constexpr (MC > 0) {
#pragma unroll
      for (int j = 0; j < MC; ++j) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
          acc[j] += a[j][k] * ng[k];
          acc[MC + j] += b[j][k] * ng[k];
        }
      }
    }
    sti<W>(iwhere + i, iw);
    st<W>(tbrk + i, tb);
  });
  // slots [0..MC) Wy'd, [MC..2MC) Ws'd, then f1, nbreak, nunb, nunbnz (sums), bkmin (min)
  block_reduce_store<NA>(acc, 2 * MC + 4, 1, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_cauchy_scan(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                        const int32_t *nbd, const T *g, iw_t *iwhere, T *tbrk, WStore<T> w,
                        int head, int col) {
  const int gr = grid_for_w(n, VecOf<T>::V, (int)sizeof(T));
  if (col == 0) {
    hipLaunchKernelGGL((cauchy_scan_kernel<T, 0, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n, x, l, u,
                       nbd, g, iwhere, tbrk, w.ws, w.wy, w.ld, w.m, head, col, q.d_part);
  } else {
    DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((cauchy_scan_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                          q.stream, n, x, l, u, nbd, g, iwhere, tbrk, w.ws, w.wy,
                                          w.ld, w.m, head, col, q.d_part));
  }
  q.launches++;
  launch_finalize(q, gr, 2 * (col == 0 ? 0 : maxc_for(col)) + 4, 1, 0);
}

// =========================== cauchy breakpoint selection =====================
__device__ __forceinline__ bool after_cursor(double t, int64_t gi, double lo_t, int64_t lo_i) {
  return t > lo_t || (t == lo_t && gi > lo_i);
}
__device__ __forceinline__ uint64_t key_of(double t) {  // t >= 0: bit pattern is monotone
  return (uint64_t)__double_as_longlong(t);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_window_kernel(int64_t n, int64_t row0,
                                                              const T *__restrict__ tbrk,
                                                              double lo_t, int64_t lo_i,
                                                              double hi_t, uint64_t *keys,
                                                              uint32_t *idx, uint32_t cap,
                                                              uint32_t *count) {
  // 4 independent 16-byte loads per lane and trip (8 rows for fp64); candidates are rare, so
  // the common trip is: loads, 8 compares, one ballot.
  constexpr int V = VecOf<T>::V, U = 4, RPT = V * U;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ngroups = (n + V - 1) / V;  // groups of V rows; the last may be partial
  const int64_t ntrips = (ngroups + nthreads * U - 1) / (nthreads * U);
  const int lane = threadIdx.x & 63;
  for (int64_t trip = 0; trip < ntrips; ++trip) {
    double tv[RPT];
    int64_t ri[RPT];
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
      const int64_t gq = (trip * U + uu) * nthreads + t0;
      const int64_t r = gq * V;
      double tmp[V];
      if (r + V <= n) {
        ld<V>(tbrk + r, tmp);
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) tmp[k] = r + k < n ? (double)tbrk[r + k] : -1.0;
      }
#pragma unroll
      for (int k = 0; k < V; ++k) {
        tv[uu * V + k] = tmp[k];
        ri[uu * V + k] = r + k;
      }
    }
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      const double t = tv[e];
      const bool pred = t >= 0.0 && t <= hi_t && after_cursor(t, row0 + ri[e], lo_t, lo_i);
      bits |= pred ? (1u << e) : 0u;
    }
    if (__ballot(bits != 0) == 0ull) continue;
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      const bool pred = (bits >> e) & 1u;
      const unsigned long long mask = __ballot(pred);
      if (mask == 0ull) continue;
      const int leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (pred) {
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (pos < cap) {
          keys[pos] = key_of(tv[e]);
          idx[pos] = (uint32_t)ri[e];
        }
      }
    }
  }
}
template <typename T>
void launch_cauchy_window(Queue &q, int64_t n, int64_t row0, const T *tbrk, double lo_t,
                          int64_t lo_i, double hi_t, uint64_t *keys, uint32_t *idx, uint32_t cap,
                          uint32_t *d_count) {
  (void)hipMemsetAsync(d_count, 0, sizeof(uint32_t), q.stream);
  const int gr = grid_for(n, VecOf<T>::V * 4);
  hipLaunchKernelGGL(cauchy_window_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, tbrk,
                     lo_t, lo_i, hi_t, keys, idx, cap, d_count);
  q.launches++;
}

// Breakpoint time of one row from its own data, exactly as the scans store it in tbrk (incl.
// the rounding to T): -1 = the row does not move, +inf = it moves without meeting a bound.
// iw is iwhere AFTER the scan's update (cauchy :1284-1291).
template <typename T>
__device__ __forceinline__ double brk_time(double xk, double lk, double uk, int nb, double gk,
                                           int iw) {
  if (iw != 0 && iw != -1) return -1.0;
  const double neggi = -gk;
  double tb = LB_INF;
  if (nb <= 2 && nb != 0 && neggi < 0.0) {
    tb = (xk - lk) / (-neggi);
  } else if (nb >= 2 && neggi > 0.0) {
    tb = (uk - xk) / neggi;
  }
  return (double)(T)tb;
}

// The window compaction without a stored tbrk: breakpoint times are recomputed per row
// (read-only pass over x, l, u, nbd, g, iwhere; the iteration's update pass then writes no
// n-vector at all, see update_scan_kernel).
template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_window_fly_kernel(
    int64_t n, int64_t row0, const T *__restrict__ x, const T *__restrict__ l,
    const T *__restrict__ u, const int32_t *__restrict__ nbd, const T *__restrict__ g,
    const iw_t *__restrict__ iwhere, double lo_t, int64_t lo_i, double hi_t, uint64_t *keys,
    uint32_t *idx, uint32_t cap, uint32_t *count) {
  const int lane = threadIdx.x & 63;
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], tv[W];
    int nb[W], iw[W];
    ldx<W, true>(x + i, xv);
    ldx<W, true>(l + i, lv);
    ldx<W, true>(u + i, uv);
    ldx<W, true>(g + i, gv);
    ldi<W>(nbd + i, nb);
    ldi<W>(iwhere + i, iw);
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      tv[k] = brk_time<T>(xv[k], lv[k], uv[k], nb[k], gv[k], iw[k]);
      const bool pred =
          tv[k] >= 0.0 && tv[k] <= hi_t && after_cursor(tv[k], row0 + i + k, lo_t, lo_i);
      bits |= pred ? (1u << k) : 0u;
    }
    if (__ballot(bits != 0) == 0ull) return;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const bool pred = (bits >> k) & 1u;
      const unsigned long long mask = __ballot(pred);
      if (mask == 0ull) continue;
      const int leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (pred) {
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (pos < cap) {
          keys[pos] = key_of(tv[k]);
          idx[pos] = (uint32_t)(i + k);
        }
      }
    }
  });
}
template <typename T>
void launch_cauchy_window_fly(Queue &q, int64_t n, int64_t row0, const T *x, const T *l, const T *u,
                              const int32_t *nbd, const T *g, const iw_t *iwhere, double lo_t,
                              int64_t lo_i, double hi_t, uint64_t *keys, uint32_t *idx, uint32_t cap,
                              uint32_t *d_count) {
  (void)hipMemsetAsync(d_count, 0, sizeof(uint32_t), q.stream);
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(cauchy_window_fly_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, x, l,
                     u, nbd, g, iwhere, lo_t, lo_i, hi_t, keys, idx, cap, d_count);
  q.launches++;
}
// iwhere update of cauchy's n-loop alone (:1284-1291), for contexts whose speculative update pass
// must leave iwhere untouched until the trial point is accepted (state mirrored at every return)
template <typename T>
__global__ __launch_bounds__(BLOCK) void iwhere_update_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const int32_t *__restrict__ nbd, const T *__restrict__ g, iw_t *iwhere) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W];
    int nb[W], iw[W];
    ld<W>(x + i, xv);
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ld<W>(g + i, gv);
    ldi<W>(nbd + i, nb);
    ldi<W>(iwhere + i, iw);
    bool changed = false;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (iw[k] != 3 && iw[k] != -1) {
        const double neggi = -gv[k];
        double tl = 0.0, tu = 0.0;
        if (nb[k] <= 2) tl = xv[k] - lv[k];
        if (nb[k] >= 2) tu = uv[k] - xv[k];
        const bool xlower = nb[k] <= 2 && tl <= 0.0;
        const bool xupper = nb[k] >= 2 && tu <= 0.0;
        const int old = iw[k];
        iw[k] = 0;
        if (xlower) {
          if (neggi <= 0.0) iw[k] = 1;
        } else if (xupper) {
          if (neggi >= 0.0) iw[k] = 2;
        } else {
          if (fabs(neggi) <= 0.0) iw[k] = -3;
        }
        changed = changed || iw[k] != old;
      }
    }
    if (__ballot(changed) != 0ull) sti<W>(iwhere + i, iw);
  });
}
template <typename T>
void launch_iwhere_update(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                          const int32_t *nbd, const T *g, iw_t *iwhere) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(iwhere_update_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, x, l, u, nbd, g,
                     iwhere);
  q.launches++;
}

// =========================== parallel GCP search, col > 0 (opt-in) ============
// SURVEY.md 8f-2.  With the breakpoints sorted, the walk's state at breakpoint k is a prefix
// sum: p_k = p_0 - sum_{j<k} d_j wbp_j, c_k = t_k p_0 - sum_{j<=k} dt_j P_j, and the f1/f2
// recurrences (:1452-1481, without the f2 >= epsmch*f2_org clamp) become two more scans once
// the quadratic forms with M are known per breakpoint.  Equal to the reference in exact
// arithmetic, not operation for operation: LBFGSB_F_PARALLEL_GCP only, single rank.
// Arrays are component-major: a[c * nbp + k], k = sorted position of the breakpoint.
template <typename T>
__global__ __launch_bounds__(BLOCK) void pgcp_gather_kernel(
    const uint32_t *__restrict__ idx, const uint64_t *__restrict__ keys, int64_t nb, int64_t nbp,
    const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const T *__restrict__ g, const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw,
    int m, int head, int col, double theta, const T *pr, const T *pd, Pend pe, double *tt,
    double *dd, double *a0, double *wb, double *uu) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    const int64_t i = idx[k];
    const double d = -(double)g[i];
    const double z = d > 0.0 ? (double)u[i] - (double)x[i] : (double)l[i] - (double)x[i];
    tt[k] = __longlong_as_double((long long)keys[k]);
    dd[k] = d;
    a0[k] = d * d - theta * d * z;
    for (int j = 0; j < col; ++j) {
      const int64_t off = (int64_t)((head - 1 + j) % m) * ldw + i;
      const bool pj = pe.on && j == col - 1;
      const double yv = pj ? pend_y<T>((double)g[i], (double)pr[i]) : (double)wy[off];
      const double sv = theta * (pj ? pend_s<T>((double)pd[i], pe.stp) : (double)ws[off]);
      wb[(int64_t)j * nbp + k] = yv;
      wb[(int64_t)(col + j) * nbp + k] = sv;
      uu[(int64_t)j * nbp + k] = d * yv;
      uu[(int64_t)(col + j) * nbp + k] = d * sv;
    }
  }
}
// q[c][k] = dt_k * P[c][k]  (P = exclusive scan of uu)
__global__ __launch_bounds__(BLOCK) void pgcp_dtp_kernel(int64_t nb, int64_t nbp, int col2,
                                                         const double *__restrict__ tt,
                                                         const double *__restrict__ pp, double *qq) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    const double dt = tt[k] - (k > 0 ? tt[k - 1] : 0.0);
    for (int c = 0; c < col2; ++c) qq[(int64_t)c * nbp + k] = dt * pp[(int64_t)c * nbp + k];
  }
}
// per breakpoint: y = M wbp, wmc = c.y, wmp = p.y, wmw = wbp.y with p = p0 - P_k (before this
// breakpoint), c = t_k p0 - SQ_k (after c += dt p);  df2 and the f2-free part of df1
__global__ __launch_bounds__(BLOCK) void pgcp_terms_kernel(
    int64_t nb, int64_t nbp, int col2, double theta, const double *__restrict__ mm /* col2 x col2 */,
    const double *__restrict__ p0, const double *__restrict__ tt, const double *__restrict__ dd,
    const double *__restrict__ a0, const double *__restrict__ wb, const double *__restrict__ pp,
    const double *__restrict__ sq, double *df2, double *a1) {
  __shared__ double sm[4 * MAXM * MAXM];
  __shared__ double sp0[2 * MAXM];
  for (int e = threadIdx.x; e < col2 * col2; e += blockDim.x) sm[e] = mm[e];
  for (int e = threadIdx.x; e < col2; e += blockDim.x) sp0[e] = p0[e];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    double w[2 * MAXM];
    for (int c = 0; c < col2; ++c) w[c] = wb[(int64_t)c * nbp + k];
    const double tk = tt[k];
    double wmc = 0.0, wmp = 0.0, wmw = 0.0;
    for (int a = 0; a < col2; ++a) {
      double y = 0.0;
      for (int b = 0; b < col2; ++b) y += sm[a + b * col2] * w[b];
      const double pa = sp0[a] - pp[(int64_t)a * nbp + k];
      const double ca = tk * sp0[a] - sq[(int64_t)a * nbp + k];
      wmc += ca * y;
      wmp += pa * y;
      wmw += w[a] * y;
    }
    const double d = dd[k];
    df2[k] = -theta * d * d + 2.0 * d * wmp - d * d * wmw;
    a1[k] = a0[k] + d * wmc;
  }
}
// df1_k = dt_k * f2_{k-1} + a1_k with f2_{k-1} = f2_0 + SF2[k-1]
__global__ __launch_bounds__(BLOCK) void pgcp_f1_kernel(int64_t nb, double f2_0,
                                                        const double *__restrict__ tt,
                                                        const double *__restrict__ sf2,
                                                        const double *__restrict__ a1, double *df1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    const double dt = tt[k] - (k > 0 ? tt[k - 1] : 0.0);
    const double f2p = f2_0 + (k > 0 ? sf2[k - 1] : 0.0);
    df1[k] = dt * f2p + a1[k];
  }
}
// first breakpoint k whose segment contains the minimiser: dtm_{k-1} < dt_k  (:1416)
__global__ __launch_bounds__(BLOCK) void pgcp_find_kernel(int64_t nb, double f1_0, double f2_0,
                                                          const double *__restrict__ tt,
                                                          const double *__restrict__ sf1,
                                                          const double *__restrict__ sf2,
                                                          double *part) {
  double acc[1] = {LB_INF};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    const double dt = tt[k] - (k > 0 ? tt[k - 1] : 0.0);
    const double f1p = f1_0 + (k > 0 ? sf1[k - 1] : 0.0);
    const double f2p = f2_0 + (k > 0 ? sf2[k - 1] : 0.0);
    const double dtm = -f1p / f2p;
    if (dtm < dt) acc[0] = fmin(acc[0], (double)k);
  }
  block_reduce_store<1>(acc, 0, 1, 0, part, MAX_BLOCKS);
}
// the state the host needs at k* (number of breakpoints crossed): out = { t_{k*-1}, f1, f2 before
// breakpoint k*, idx of breakpoint k*-1, then P[c][k*] (c < col2), then SQ[c][k*-1] }
__global__ void pgcp_pick_kernel(int64_t ks, int64_t nb, int64_t nbp, int col2, double f1_0,
                                 double f2_0, const double *__restrict__ tt,
                                 const double *__restrict__ sf1, const double *__restrict__ sf2,
                                 const double *__restrict__ pp, const double *__restrict__ uu_last,
                                 const double *__restrict__ sq, const uint32_t *__restrict__ idx,
                                 double *out) {
  const int c = threadIdx.x;
  if (c == 0) {
    out[0] = ks > 0 ? tt[ks - 1] : 0.0;
    out[1] = f1_0 + (ks > 0 ? sf1[ks - 1] : 0.0);
    out[2] = f2_0 + (ks > 0 ? sf2[ks - 1] : 0.0);
    out[3] = ks > 0 ? (double)idx[ks - 1] : -1.0;
  }
  if (c < col2) {
    // exclusive prefix at ks; for ks == nb it is the last exclusive prefix plus the last term,
    // which the caller kept in uu_last (the scan ran in place)
    out[4 + c] = ks < nb ? pp[(int64_t)c * nbp + ks] : pp[(int64_t)c * nbp + nb - 1] + uu_last[c];
    out[4 + col2 + c] = ks > 0 ? sq[(int64_t)c * nbp + ks - 1] : 0.0;
  }
}
// uu_last[c] = uu[c][nb-1] before the in-place exclusive scan
__global__ void pgcp_last_kernel(int64_t nb, int64_t nbp, int col2, const double *__restrict__ uu,
                                 double *uu_last) {
  const int c = threadIdx.x;
  if (c < col2) uu_last[c] = uu[(int64_t)c * nbp + nb - 1];
}

size_t scan_temp_bytes(size_t count) {
  size_t b1 = 0, b2 = 0;
  (void)rocprim::inclusive_scan(nullptr, b1, (const double *)nullptr, (double *)nullptr, count,
                                rocprim::plus<double>(), (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, b2, (const double *)nullptr, (double *)nullptr, 0.0, count,
                                rocprim::plus<double>(), (hipStream_t)0);
  return b1 > b2 ? b1 : b2;
}
void launch_scan(Queue &q, void *d_temp, size_t temp_bytes, const double *in, double *out,
                 size_t count, int exclusive) {
  if (exclusive)
    (void)rocprim::exclusive_scan(d_temp, temp_bytes, in, out, 0.0, count, rocprim::plus<double>(),
                                  q.stream);
  else
    (void)rocprim::inclusive_scan(d_temp, temp_bytes, in, out, count, rocprim::plus<double>(),
                                  q.stream);
  q.launches++;
}
template <typename T>
void launch_pgcp_gather(Queue &q, const uint32_t *idx, const uint64_t *keys, int64_t nb, int64_t nbp,
                        const T *x, const T *l, const T *u, const T *g, WStore<T> w, int head, int col,
                        double theta, const T *pr, const T *pd, Pend pe, double *tt, double *dd,
                        double *a0, double *wb, double *uu) {
  const int gr = grid_for(nb, 1);
  hipLaunchKernelGGL(pgcp_gather_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, idx, keys, nb, nbp, x,
                     l, u, g, w.ws, w.wy, w.ld, w.m, head, col, theta, pr, pd, pe, tt, dd, a0, wb, uu);
  q.launches++;
}
void launch_pgcp_last(Queue &q, int64_t nb, int64_t nbp, int col2, const double *uu, double *uu_last) {
  hipLaunchKernelGGL(pgcp_last_kernel, dim3(1), dim3(64), 0, q.stream, nb, nbp, col2, uu, uu_last);
  q.launches++;
}
void launch_pgcp_dtp(Queue &q, int64_t nb, int64_t nbp, int col2, const double *tt, const double *pp,
                     double *qq) {
  hipLaunchKernelGGL(pgcp_dtp_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, nbp, col2,
                     tt, pp, qq);
  q.launches++;
}
void launch_pgcp_terms(Queue &q, int64_t nb, int64_t nbp, int col2, double theta, const double *mm,
                       const double *p0, const double *tt, const double *dd, const double *a0,
                       const double *wb, const double *pp, const double *sq, double *df2, double *a1) {
  hipLaunchKernelGGL(pgcp_terms_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, nbp, col2,
                     theta, mm, p0, tt, dd, a0, wb, pp, sq, df2, a1);
  q.launches++;
}
void launch_pgcp_f1(Queue &q, int64_t nb, double f2_0, const double *tt, const double *sf2,
                    const double *a1, double *df1) {
  hipLaunchKernelGGL(pgcp_f1_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, f2_0, tt, sf2,
                     a1, df1);
  q.launches++;
}
void launch_pgcp_find(Queue &q, int64_t nb, double f1_0, double f2_0, const double *tt,
                      const double *sf1, const double *sf2) {
  const int gr = grid_for(nb, 1);
  hipLaunchKernelGGL(pgcp_find_kernel, dim3(gr), dim3(BLOCK), 0, q.stream, nb, f1_0, f2_0, tt, sf1, sf2,
                     q.d_part);
  q.launches++;
  launch_finalize(q, gr, 0, 1, 0);
}
void launch_pgcp_pick(Queue &q, int64_t ks, int64_t nb, int64_t nbp, int col2, double f1_0, double f2_0,
                      const double *tt, const double *sf1, const double *sf2, const double *pp,
                      const double *uu_last, const double *sq, const uint32_t *idx, double *out) {
  hipLaunchKernelGGL(pgcp_pick_kernel, dim3(1), dim3(64), 0, q.stream, ks, nb, nbp, col2, f1_0, f2_0, tt,
                     sf1, sf2, pp, uu_last, sq, idx, out);
  q.launches++;
}

// tbrk as a vector, for the paths that want one (full sort, cursor-based cauchy_finish)
template <typename T>
__global__ __launch_bounds__(BLOCK) void tbrk_fill_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const int32_t *__restrict__ nbd, const T *__restrict__ g, const iw_t *__restrict__ iwhere,
    T *tbrk) {
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], tv[W];
    int nb[W], iw[W];
    ld<W>(x + i, xv);
    ld<W>(l + i, lv);
    ld<W>(u + i, uv);
    ld<W>(g + i, gv);
    ldi<W>(nbd + i, nb);
    ldi<W>(iwhere + i, iw);
#pragma unroll
    for (int k = 0; k < W; ++k) tv[k] = brk_time<T>(xv[k], lv[k], uv[k], nb[k], gv[k], iw[k]);
    st<W>(tbrk + i, tv);
  });
}
template <typename T>
void launch_tbrk_fill(Queue &q, int64_t n, const T *x, const T *l, const T *u, const int32_t *nbd,
                      const T *g, const iw_t *iwhere, T *tbrk) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(tbrk_fill_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, x, l, u, nbd, g,
                     iwhere, tbrk);
  q.launches++;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_allkeys_kernel(int64_t n, int64_t row0,
                                                               const T *__restrict__ tbrk,
                                                               double lo_t, int64_t lo_i,
                                                               uint64_t *keys, uint32_t *idx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double t = (double)tbrk[i];
    const bool pred = t >= 0.0 && t < LB_INF && after_cursor(t, row0 + i, lo_t, lo_i);
    keys[i] = pred ? key_of(t) : ~0ull;
    idx[i] = (uint32_t)i;
  }
}
template <typename T>
void launch_cauchy_allkeys(Queue &q, int64_t n, int64_t row0, const T *tbrk, double lo_t,
                           int64_t lo_i, uint64_t *keys, uint32_t *idx) {
  const int gr = grid_for(n, 1);
  hipLaunchKernelGGL(cauchy_allkeys_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, tbrk,
                     lo_t, lo_i, keys, idx);
  q.launches++;
}

size_t sort_pairs_temp_bytes(size_t count) {
  size_t b1 = 0, b2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b1, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, count, 0, 64,
                                  (hipStream_t)0);
  (void)rocprim::radix_sort_pairs(nullptr, b2, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (const uint64_t *)nullptr, (uint64_t *)nullptr, count, 0, 32,
                                  (hipStream_t)0);
  return b1 > b2 ? b1 : b2;
}
void launch_sort_by_idx(Queue &q, void *d_temp, size_t temp_bytes, const uint32_t *idx_in,
                        uint32_t *idx_out, const uint64_t *keys_in, uint64_t *keys_out,
                        size_t count) {
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, idx_in, idx_out, keys_in, keys_out, count, 0,
                                  32, q.stream);
  q.launches++;
}
void launch_sort_pairs(Queue &q, void *d_temp, size_t temp_bytes, const uint64_t *keys_in,
                       uint64_t *keys_out, const uint32_t *idx_in, uint32_t *idx_out,
                       size_t count) {
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, count, 0,
                                  64, q.stream);
  q.launches++;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_gather_kernel(
    const uint32_t *__restrict__ idx, const uint64_t *__restrict__ keys, uint32_t cnt,
    int64_t row0, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const T *__restrict__ g, const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw,
    int m, int head, int col, const T *pr, const T *pd, Pend pe, double *rec) {
  const int rl = 2 * col + 4;
  const int64_t total = (int64_t)cnt * rl;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = (uint32_t)(q / rl);
    const int f = (int)(q % rl);
    const int64_t i = idx[k];
    double v;
    if (f == 0) {
      v = __longlong_as_double((long long)keys[k]);  // the breakpoint time IS the sort key
    } else if (f == 1) {
      v = (double)(row0 + i);
    } else if (f == 2) {
      v = -(double)g[i];
    } else if (f == 3) {
      const double d = -(double)g[i];
      v = d > 0.0 ? (double)u[i] - (double)x[i] : (double)l[i] - (double)x[i];
    } else if (f < 4 + col) {
      v = (pe.on && f - 4 == col - 1) ? pend_y<T>((double)g[i], (double)pr[i])
                                      : (double)wy[(int64_t)((head - 1 + (f - 4)) % m) * ldw + i];
    } else {
      v = (pe.on && f - 4 - col == col - 1)
              ? pend_s<T>((double)pd[i], pe.stp)
              : (double)ws[(int64_t)((head - 1 + (f - 4 - col)) % m) * ldw + i];
    }
    rec[q] = v;
  }
}
// Fast path of the window fetch: the candidate count stays on the device.  Gathers the
// records of the first min(*d_count, cap) candidates (unordered, as the window kernel appended
// them) and writes the header {count, 0} in front, so ONE host sync delivers everything a
// short walk needs; the host orders the few records itself.
template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_gather_dyn_kernel(
    const uint32_t *__restrict__ idx, const uint64_t *__restrict__ keys,
    const uint32_t *__restrict__ d_count, uint32_t cap, int64_t row0, const T *__restrict__ x,
    const T *__restrict__ l, const T *__restrict__ u, const T *__restrict__ g,
    const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw, int m, int head, int col,
    const T *pr, const T *pd, Pend pe, double *msg) {
  const uint32_t total_cnt = *d_count;
  const uint32_t cnt = total_cnt < cap ? total_cnt : cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    msg[0] = (double)total_cnt;
    msg[1] = 0.0;
  }
  double *rec = msg + 2;
  const int rl = 2 * col + 4;
  const int64_t total = (int64_t)cnt * rl;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = (uint32_t)(q / rl);
    const int f = (int)(q % rl);
    const int64_t i = idx[k];
    double v;
    if (f == 0) {
      v = __longlong_as_double((long long)keys[k]);  // the breakpoint time IS the sort key
    } else if (f == 1) {
      v = (double)(row0 + i);
    } else if (f == 2) {
      v = -(double)g[i];
    } else if (f == 3) {
      const double d = -(double)g[i];
      v = d > 0.0 ? (double)u[i] - (double)x[i] : (double)l[i] - (double)x[i];
    } else if (f < 4 + col) {
      v = (pe.on && f - 4 == col - 1) ? pend_y<T>((double)g[i], (double)pr[i])
                                      : (double)wy[(int64_t)((head - 1 + (f - 4)) % m) * ldw + i];
    } else {
      v = (pe.on && f - 4 - col == col - 1)
              ? pend_s<T>((double)pd[i], pe.stp)
              : (double)ws[(int64_t)((head - 1 + (f - 4 - col)) % m) * ldw + i];
    }
    rec[q] = v;
  }
}
template <typename T>
void launch_cauchy_gather_dyn(Queue &q, const uint32_t *idx, const uint64_t *keys,
                              const uint32_t *d_count, uint32_t cap, int64_t row0, const T *x,
                              const T *l, const T *u, const T *g, WStore<T> w, int head, int col,
                              const T *pr, const T *pd, Pend pe, double *msg) {
  const int64_t total = (int64_t)cap * (2 * col + 4);
  int gr = (int)((total + BLOCK - 1) / BLOCK);
  if (gr > 64) gr = 64;
  hipLaunchKernelGGL(cauchy_gather_dyn_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, idx, keys,
                     d_count, cap, row0, x, l, u, g, w.ws, w.wy, w.ld, w.m, head, col, pr, pd, pe, msg);
  q.launches++;
}

template <typename T>
void launch_cauchy_gather(Queue &q, const uint32_t *idx, const uint64_t *keys, uint32_t cnt,
                          int64_t row0, const T *x, const T *l, const T *u, const T *g, WStore<T> w,
                          int head, int col, const T *pr, const T *pd, Pend pe, double *rec) {
  if (cnt == 0) return;
  const int64_t total = (int64_t)cnt * (2 * col + 4);
  int gr = (int)((total + BLOCK - 1) / BLOCK);
  if (gr > MAX_BLOCKS) gr = MAX_BLOCKS;
  hipLaunchKernelGGL(cauchy_gather_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, idx, keys, cnt,
                     row0, x, l, u, g, w.ws, w.wy, w.ld, w.m, head, col, pr, pd, pe, rec);
  q.launches++;
}

// COUNT: also return the number of rows fixed (closed-form GCP, where no walk counted them)
template <typename T, bool COUNT>
__global__ __launch_bounds__(BLOCK) void cauchy_finish_kernel(
    int64_t n, int64_t row0, const T *__restrict__ x, const T *__restrict__ l,
    const T *__restrict__ u, const T *__restrict__ g, const T *__restrict__ tbrk,
    iw_t *iwhere, T *xcp, double tsum, double last_t, int64_t last_i, double *part) {
  double acc[1] = {0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], gv[W], tb[W], out[W];
    ld<W>(x + i, xv);
    ld<W>(g + i, gv);
    ld<W>(tbrk + i, tb);
    // which rows were fixed by the walk?  Usually none or few: the bounds and iwhere are only
    // touched by the waves that need them (wave-uniform branch)
    bool done[W];
    bool any = false;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      done[k] = tb[k] >= 0.0 &&
                (tb[k] < last_t || (tb[k] == last_t && (row0 + i + k) <= last_i));
      any = any || done[k];
      if (COUNT && done[k]) acc[0] += 1.0;
    }
    const bool wave_any = __ballot(any) != 0ull;
    double lv[W], uv[W];
    int iw[W];
    if (wave_any) {
      ld<W>(l + i, lv);
      ld<W>(u + i, uv);
      ldi<W>(iwhere + i, iw);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      out[k] = xv[k];
      if (tb[k] >= 0.0) {
        const double d = -gv[k];
        if (done[k]) {
          if (d > 0.0) {
            out[k] = uv[k];
            iw[k] = 2;
          } else {
            out[k] = lv[k];
            iw[k] = 1;
          }
        } else if (tsum != 0.0) {
          out[k] = xv[k] + tsum * d;
        }
      }
    }
    st<W>(xcp + i, out);
    if (wave_any) sti<W>(iwhere + i, iw);
  });
  if constexpr (COUNT) block_reduce_store<1>(acc, 1, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_cauchy_finish(Queue &q, int64_t n, int64_t row0, const T *x, const T *l, const T *u,
                          const T *g, const T *tbrk, iw_t *iwhere, T *xcp, double tsum,
                          double last_t, int64_t last_i, int count) {
  const int gr = grid_for(n, VecOf<T>::V);
  if (count) {
    hipLaunchKernelGGL((cauchy_finish_kernel<T, true>), dim3(gr), dim3(BLOCK), 0, q.stream, n, row0,
                       x, l, u, g, tbrk, iwhere, xcp, tsum, last_t, last_i, q.d_part);
    q.launches++;
    launch_finalize(q, gr, 1, 0, 0);
  } else {
    hipLaunchKernelGGL((cauchy_finish_kernel<T, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n, row0,
                       x, l, u, g, tbrk, iwhere, xcp, tsum, last_t, last_i, q.d_part);
    q.launches++;
  }
}

// rows fixed by a short walk, as a list: entry = global row * 2 + (1 if fixed at the upper bound)
__global__ void cauchy_fix_kernel(const int64_t *__restrict__ list, int count, int64_t row0,
                                  int64_t n, iw_t *iwhere) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int64_t gi = list[k] >> 1;
  if (gi >= row0 && gi < row0 + n) iwhere[gi - row0] = (list[k] & 1) ? 2 : 1;
}
void launch_cauchy_fix(Queue &q, const int64_t *list, int count, int64_t row0, int64_t n,
                       iw_t *iwhere) {
  hipLaunchKernelGGL(cauchy_fix_kernel, dim3((count + 255) / 256), dim3(256), 0, q.stream, list, count,
                     row0, n, iwhere);
  q.launches++;
}

// =========================== freev (:1980-2059) ==============================
__global__ __launch_bounds__(BLOCK) void freev_count_kernel(int64_t n,
                                                            const iw_t *__restrict__ iwhere,
                                                            int8_t *wasfree, double *part,
                                                            uint32_t *chg, uint32_t chg_cap,
                                                            uint32_t *chg_count) {
  // rows whose status changed are collected per workgroup in LDS and appended to the global
  // list with ONE global atomic per flush (a same-address atomic per row would serialise:
  // 1e5 changes x ~12 ns)
  constexpr int LCAP = 2048;
  __shared__ uint32_t lbuf[LCAP];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  double acc[3] = {0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t ntrip = (n + stride - 1) / stride;  // uniform trip count (barriers inside)
  for (int64_t trip = 0; trip < ntrip; ++trip) {
    const int64_t i = trip * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
      const bool fr = iwhere[i] <= 0;
      const bool was = wasfree[i] != 0;
      if (fr) acc[0] += 1.0;
      if (fr && !was) acc[1] += 1.0;
      if (!fr && was) acc[2] += 1.0;
      if (chg && fr != was) {
        const uint32_t pos = atomicAdd(&lcount, 1u);  // LDS atomic; pos < LCAP by the flush rule
        lbuf[pos] = (uint32_t)i | (fr ? 0u : 0x80000000u);
      }
      if (fr != was) wasfree[i] = fr ? 1 : 0;  // (few rows: keeps the pass that follows free of
                                               //  drained store traffic)
    }
    if (chg) {
      __syncthreads();
      const uint32_t cnt = lcount;
      if (cnt > LCAP - BLOCK || trip == ntrip - 1) {  // uniform: flush
        if (threadIdx.x == 0) gbase = cnt ? atomicAdd(chg_count, cnt) : 0u;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt; k += BLOCK)
          if (gbase + k < chg_cap) chg[gbase + k] = lbuf[k];
        __syncthreads();
        if (threadIdx.x == 0) lcount = 0;
        __syncthreads();
      }
    }
  }
  block_reduce_store<3>(acc, 3, 0, 0, part, MAX_BLOCKS);
}
void launch_freev_count(Queue &q, int64_t n, const iw_t *iwhere, int8_t *wasfree, uint32_t *chg,
                        uint32_t chg_cap, uint32_t *chg_count) {
  const int gr = grid_for(n, 1);
  if (chg) (void)hipMemsetAsync(chg_count, 0, sizeof(uint32_t), q.stream);
  hipLaunchKernelGGL(freev_count_kernel, dim3(gr), dim3(BLOCK), 0, q.stream, n, iwhere, wasfree,
                     q.d_part, chg, chg_cap, chg_count);
  q.launches++;
  launch_finalize(q, gr, 3, 0, 0);
}

// ordered stream compaction reproducing the reference's list orders exactly:
//   Index : free variables ascending from the front, active ascending from the back
//   Indx2 : entering in DESCENDING variable order from the front (the reference walks the
//           old active list, which is stored back to front), leaving ascending from the back.
constexpr int LIST_ITEMS = 4;
constexpr int LIST_CHUNK = BLOCK * LIST_ITEMS;

__device__ __forceinline__ void list_flags(int64_t i, int64_t n, const iw_t *iwhere,
                                           const int8_t *prev, int do_el, int &fr, int &en,
                                           int &lv) {
  fr = en = lv = 0;
  if (i < n) {
    fr = iwhere[i] <= 0;
    if (do_el) {
      const int was = prev[i] != 0;
      en = fr && !was;
      lv = !fr && was;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void list_count_kernel(int64_t n, const iw_t *iwhere,
                                                           const int8_t *prev, int do_el,
                                                           int32_t *tmp) {
  __shared__ int s[3];
  if (threadIdx.x < 3) s[threadIdx.x] = 0;
  __syncthreads();
  int c0 = 0, c1 = 0, c2 = 0;
  for (int k = 0; k < LIST_ITEMS; ++k) {
    int fr, en, lv;
    list_flags((int64_t)blockIdx.x * LIST_CHUNK + threadIdx.x * LIST_ITEMS + k, n, iwhere, prev,
               do_el, fr, en, lv);
    c0 += fr;
    c1 += en;
    c2 += lv;
  }
  atomicAdd(&s[0], c0);
  atomicAdd(&s[1], c1);
  atomicAdd(&s[2], c2);
  __syncthreads();
  if (threadIdx.x < 3) tmp[3 * blockIdx.x + threadIdx.x] = s[threadIdx.x];
}
// exclusive scan of the per-chunk counts (single workgroup); totals in tmp[3*nch ..]
__global__ __launch_bounds__(BLOCK) void list_scan_kernel(int nch, int32_t *tmp) {
  __shared__ int tot[3][BLOCK];
  const int per = (nch + BLOCK - 1) / BLOCK;
  const int b0 = threadIdx.x * per, b1 = min(nch, b0 + per);
  int c[3] = {0, 0, 0};
  for (int b = b0; b < b1; ++b)
    for (int k = 0; k < 3; ++k) c[k] += tmp[3 * b + k];
  for (int k = 0; k < 3; ++k) tot[k][threadIdx.x] = c[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 3; ++k) {
      int run = 0;
      for (int t = 0; t < BLOCK; ++t) {
        const int v = tot[k][t];
        tot[k][t] = run;
        run += v;
      }
      tmp[3 * nch + k] = run;
    }
  }
  __syncthreads();
  int run[3] = {tot[0][threadIdx.x], tot[1][threadIdx.x], tot[2][threadIdx.x]};
  for (int b = b0; b < b1; ++b)
    for (int k = 0; k < 3; ++k) {
      const int v = tmp[3 * b + k];
      tmp[3 * b + k] = run[k];
      run[k] += v;
    }
}
__global__ __launch_bounds__(BLOCK) void list_write_kernel(int64_t n, const iw_t *iwhere,
                                                           const int8_t *prev, int do_el,
                                                           const int32_t *tmp, int nch,
                                                           int32_t *index, int32_t *indx2) {
  __shared__ int sc[3][BLOCK];
  int fr[LIST_ITEMS], en[LIST_ITEMS], lv[LIST_ITEMS];
  int c[3] = {0, 0, 0};
  const int64_t i0 = (int64_t)blockIdx.x * LIST_CHUNK + threadIdx.x * LIST_ITEMS;
  for (int k = 0; k < LIST_ITEMS; ++k) {
    list_flags(i0 + k, n, iwhere, prev, do_el, fr[k], en[k], lv[k]);
    c[0] += fr[k];
    c[1] += en[k];
    c[2] += lv[k];
  }
  for (int k = 0; k < 3; ++k) sc[k][threadIdx.x] = c[k];
  __syncthreads();
  if (threadIdx.x < 3) {
    int run = 0;
    for (int t = 0; t < BLOCK; ++t) {
      const int v = sc[threadIdx.x][t];
      sc[threadIdx.x][t] = run;
      run += v;
    }
  }
  __syncthreads();
  int64_t pf = (int64_t)tmp[3 * blockIdx.x + 0] + sc[0][threadIdx.x];
  int64_t pe = (int64_t)tmp[3 * blockIdx.x + 1] + sc[1][threadIdx.x];
  int64_t pl = (int64_t)tmp[3 * blockIdx.x + 2] + sc[2][threadIdx.x];
  const int64_t nenter = tmp[3 * nch + 1];
  for (int k = 0; k < LIST_ITEMS; ++k) {
    const int64_t i = i0 + k;
    if (i >= n) break;
    const int32_t var = (int32_t)(i + 1);
    if (fr[k]) {
      index[pf] = var;
      pf++;
    } else {
      const int64_t ar = i - pf;  // actives before i
      index[n - 1 - ar] = var;
    }
    if (en[k]) {
      indx2[nenter - 1 - pe] = var;
      pe++;
    }
    if (lv[k]) {
      indx2[n - 1 - pl] = var;
      pl++;
    }
  }
}
void launch_freev_lists(Queue &q, int64_t n, const iw_t *iwhere, const int8_t *prevfree,
                        int do_enterleave, int32_t *index, int32_t *indx2, int32_t *scan_tmp) {
  const int nch = (int)((n + LIST_CHUNK - 1) / LIST_CHUNK);
  hipLaunchKernelGGL(list_count_kernel, dim3(nch), dim3(BLOCK), 0, q.stream, n, iwhere, prevfree,
                     do_enterleave, scan_tmp);
  hipLaunchKernelGGL(list_scan_kernel, dim3(1), dim3(BLOCK), 0, q.stream, nch, scan_tmp);
  hipLaunchKernelGGL(list_write_kernel, dim3(nch), dim3(BLOCK), 0, q.stream, n, iwhere, prevfree,
                     do_enterleave, scan_tmp, nch, index, indx2);
  q.launches += 3;
}

// =========================== formk inner products ============================
// From-scratch masked Gram of [Wy Ws] (reference keeps wn1 incrementally,
// :1735-1851; same sums, same row sets, no cliff when many variables change
// status).  Row tiles are staged in LDS once and every needed product pair reads
// them from there; each output entry is owned by exactly one lane of the
// workgroup, so no cross-lane reduction is needed.
template <int MC>
struct GramCfg {
  static constexpr int R = MC <= 20 ? 128 : 64;     // rows per tile
  static constexpr int RS = 2 * MC + 1;             // LDS row stride (odd: spreads banks)
  static constexpr int E = 2 * MC * MC + MC;        // outputs at col == MC
  static constexpr int NE = (E + BLOCK - 1) / BLOCK;  // outputs per lane
};
template <typename T, int MC>
__global__ __launch_bounds__(BLOCK) void formk_gram_kernel(int64_t n, const T *__restrict__ ws,
                                                           const T *__restrict__ wy, int64_t ldw,
                                                           int m, int head, int col,
                                                           const iw_t *__restrict__ iwhere,
                                                           double *gpart) {
  using C = GramCfg<MC>;
  __shared__ double tile[C::R * C::RS];
  __shared__ int flag[C::R];
  const int tri = col * (col + 1) / 2;
  const int E = 2 * col * col + col;
  // which (column a, column b, row set) this lane owns
  int ca[C::NE], cb[C::NE], want[C::NE];
  double acc[C::NE];
#pragma unroll
  for (int s = 0; s < C::NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    acc[s] = 0.0;
    ca[s] = cb[s] = 0;
    want[s] = 2;  // matches no row
    if (e < E) {
      if (e < 2 * tri) {
        const int ee = e < tri ? e : e - tri;
        int i = (int)((sqrt(8.0 * ee + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > ee) --i;
        while ((i + 1) * (i + 2) / 2 <= ee) ++i;
        const int j = ee - i * (i + 1) / 2;
        if (e < tri) {
          ca[s] = i;
          cb[s] = j;
          want[s] = 1;  // free rows: Wy_i . Wy_j
        } else {
          ca[s] = col + i;
          cb[s] = col + j;
          want[s] = 0;  // active rows: Ws_i . Ws_j
        }
      } else {
        const int ee = e - 2 * tri;
        const int i = ee / col, j = ee % col;
        ca[s] = col + i;  // Ws_i
        cb[s] = j;        // Wy_j
        want[s] = i > j ? 0 : 1;
      }
    }
  }
  const int64_t ntiles = (n + C::R - 1) / C::R;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t r0 = t * C::R;
    __syncthreads();
    for (int qd = threadIdx.x; qd < 2 * col * C::R; qd += BLOCK) {
      const int c = qd / C::R, r = qd % C::R;
      const int64_t row = r0 + r;
      double v = 0.0;
      if (row < n) {
        const int jj = c < col ? c : c - col;
        const int64_t off = (int64_t)((head - 1 + jj) % m) * ldw + row;
        v = c < col ? (double)wy[off] : (double)ws[off];
      }
      tile[r * C::RS + c] = v;
    }
    for (int r = threadIdx.x; r < C::R; r += BLOCK) {
      const int64_t row = r0 + r;
      flag[r] = row < n ? (iwhere[row] <= 0 ? 1 : 0) : 3;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < C::R; ++r) {
      const int f = flag[r];
#pragma unroll
      for (int s = 0; s < C::NE; ++s) {
        const double a = tile[r * C::RS + ca[s]];
        const double b = tile[r * C::RS + cb[s]];
        if (f == want[s]) acc[s] += a * b;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < C::NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    if (e < E) gpart[(size_t)e * GRAM_BLOCKS + blockIdx.x] = acc[s];
  }
}
// Row-parallel variant for col <= 10 (the benchmark's m = 10).  A 512-thread workgroup
// (8 waves) takes a 128-row slab: each wave loads 1/8 of the 2*MC columns once (16 B per
// lane, coalesced) into a double-buffered LDS slab (conflict-free 16-byte slots, ONE barrier
// per slab, the next slab's global loads in flight during the math), and each wave owns one
// eighth of the outputs in registers -- matrix = wave/2: Y'ZZ'Y (free rows), S'AA'S (active
// rows), R_z (free, i<=j), L_a (active, i>j); half = wave%2 splits the outer index at H.
// <= 28 accumulators per lane keep it under 128 VGPRs: two workgroups (16 waves) per CU.
// HBM traffic is exactly one pass over W plus iwhere.
template <int MC>
struct GramRows {
  static constexpr int H = MC == 10 ? 7 : (MC + 1) / 2 + 1;  // outer-index split
  static constexpr int NACC = H * (H + 1) / 2 > (MC - H) * (MC + H + 1) / 2
                                  ? H * (H + 1) / 2
                                  : (MC - H) * (MC + H + 1) / 2;
  static constexpr int ROWS = 128;  // 64 lanes x 2 rows
  static constexpr int NW = 8;
};

// accumulate one slab for role (MT, HALF); a = pointer to the slab [2*MC][64] of double2
template <int MC, int MT, int HALF>
__device__ __forceinline__ void gram_role(const double2 (*__restrict__ sl)[64], int lane, double m0,
                                          double m1, double (&acc)[GramRows<MC>::NACC]) {
  constexpr int H = GramRows<MC>::H;
  constexpr int LO = HALF == 0 ? 0 : H, HI = HALF == 0 ? H : MC;
  // inner operands are re-read from LDS (cheap: the LDS pipe is otherwise idle)
  if constexpr (MT == 0 || MT == 1) {
    constexpr int C0 = MT == 0 ? 0 : MC;  // Y block or S block
    int k = 0;
#pragma unroll
    for (int i = LO; i < HI; ++i) {
      const double2 ai = sl[C0 + i][lane];
      const double ax = ai.x * m0, ay = ai.y * m1;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const double2 aj = sl[C0 + j][lane];
        acc[k++] += ax * aj.x + ay * aj.y;
      }
    }
  } else if constexpr (MT == 2) {  // R_z: Ws_i . Wy_j, free rows, i <= j, outer j in [LO,HI)
    int k = 0;
#pragma unroll
    for (int j = LO; j < HI; ++j) {
      const double2 y = sl[j][lane];
      const double yx = y.x * m0, yy = y.y * m1;
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        const double2 sv = sl[MC + i][lane];
        acc[k++] += sv.x * yx + sv.y * yy;
      }
    }
  } else {  // L_a: Ws_i . Wy_j, active rows, i > j, outer i in [max(LO,1),HI)
    int k = 0;
#pragma unroll
    for (int i = (LO < 1 ? 1 : LO); i < HI; ++i) {
      const double2 sv = sl[MC + i][lane];
      const double sx = sv.x * m0, sy = sv.y * m1;
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const double2 y = sl[j][lane];
        acc[k++] += sx * y.x + sy * y.y;
      }
    }
  }
}
// write one role's outputs (same enumeration order as gram_role)
template <int MC, int MT, int HALF>
__device__ __forceinline__ void gram_store(const double (&acc)[GramRows<MC>::NACC], int lane, int col,
                                           double *gpart) {
  constexpr int H = GramRows<MC>::H;
  constexpr int LO = HALF == 0 ? 0 : H, HI = HALF == 0 ? H : MC;
  const int tri = col * (col + 1) / 2;
  int k = 0;
  if constexpr (MT == 0 || MT == 1) {
#pragma unroll
    for (int i = LO; i < HI; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const double v = wave_sum(acc[k++]);
        if (lane == 0 && i < col)
          gpart[(size_t)((MT == 0 ? 0 : tri) + i * (i + 1) / 2 + j) * GRAM_BLOCKS + blockIdx.x] = v;
      }
  } else if constexpr (MT == 2) {
#pragma unroll
    for (int j = LO; j < HI; ++j)
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        const double v = wave_sum(acc[k++]);
        if (lane == 0 && j < col) gpart[(size_t)(2 * tri + i * col + j) * GRAM_BLOCKS + blockIdx.x] = v;
      }
  } else {
#pragma unroll
    for (int i = (LO < 1 ? 1 : LO); i < HI; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const double v = wave_sum(acc[k++]);
        if (lane == 0 && i < col) gpart[(size_t)(2 * tri + i * col + j) * GRAM_BLOCKS + blockIdx.x] = v;
      }
  }
}

template <typename T, int MC>
__global__ __launch_bounds__(512) void formk_gram_rows_kernel(
    int64_t n, const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw, int m, int head,
    int col, const iw_t *__restrict__ iwhere, double *gpart) {
  using G = GramRows<MC>;
  constexpr int NC = 2 * MC;                      // columns: [0,MC) = Wy, [MC,2MC) = Ws
  constexpr int PER = (NC + G::NW - 1) / G::NW;   // columns loaded per wave
  __shared__ double2 slab[2][NC][64];
  __shared__ int2 fl[2][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform role
  double acc[G::NACC];
#pragma unroll
  for (int k = 0; k < G::NACC; ++k) acc[k] = 0.0;

  const int64_t nslab = (n + G::ROWS - 1) / G::ROWS;
  // two register stages: while slab t is computed from LDS, slabs t+1 and t+2 are in flight
  double2 stA[PER], stB[PER];
  int2 fA = make_int2(3, 3), fB = make_int2(3, 3);
  auto issue = [&](int64_t sl, double2(&stage)[PER], int2 &fstage) {
    const int64_t r0 = sl * G::ROWS + 2 * lane;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = w + G::NW * q;
      double2 v = make_double2(0.0, 0.0);
      if (sl < nslab && c < NC && r0 < n) {
        const int j = c < MC ? c : c - MC;
        const T *base = (c < MC ? wy : ws) + col_off(j, col, head, m, ldw) + r0;
        if (r0 + 1 < n) {
          double t2[2];
          ld<2>(base, t2);
          v = make_double2(t2[0], t2[1]);
        } else {
          v.x = (double)base[0];
        }
      }
      stage[q] = v;
    }
    if (w == G::NW - 1) {
      fstage = make_int2(3, 3);
      if (sl < nslab && r0 < n) fstage.x = iwhere[r0] <= 0 ? 1 : 0;
      if (sl < nslab && r0 + 1 < n) fstage.y = iwhere[r0 + 1] <= 0 ? 1 : 0;
    }
  };
  auto put = [&](int buf, const double2(&stage)[PER], const int2 &fstage) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = w + G::NW * q;
      if (c < NC) slab[buf][c][lane] = stage[q];
    }
    if (w == G::NW - 1) fl[buf][lane] = fstage;
  };
  auto math = [&](int buf) {
    const int2 f = fl[buf][lane];
    const int mt = w >> 1;
    const int want = (mt == 0 || mt == 2) ? 1 : 0;  // free rows for Y'ZZ'Y and R_z
    const double m0 = f.x == want ? 1.0 : 0.0, m1 = f.y == want ? 1.0 : 0.0;
    switch (w) {
      case 0: gram_role<MC, 0, 0>(slab[buf], lane, m0, m1, acc); break;
      case 1: gram_role<MC, 0, 1>(slab[buf], lane, m0, m1, acc); break;
      case 2: gram_role<MC, 1, 0>(slab[buf], lane, m0, m1, acc); break;
      case 3: gram_role<MC, 1, 1>(slab[buf], lane, m0, m1, acc); break;
      case 4: gram_role<MC, 2, 0>(slab[buf], lane, m0, m1, acc); break;
      case 5: gram_role<MC, 2, 1>(slab[buf], lane, m0, m1, acc); break;
      case 6: gram_role<MC, 3, 0>(slab[buf], lane, m0, m1, acc); break;
      default: gram_role<MC, 3, 1>(slab[buf], lane, m0, m1, acc); break;
    }
  };
  const int64_t g = gridDim.x;
  issue(blockIdx.x, stA, fA);
  issue(blockIdx.x + g, stB, fB);
  for (int64_t sl = blockIdx.x; sl < nslab; sl += 2 * g) {
    put(0, stA, fA);
    __syncthreads();
    issue(sl + 2 * g, stA, fA);
    math(0);
    if (sl + g < nslab) {  // uniform over the workgroup
      put(1, stB, fB);
      __syncthreads();
      issue(sl + 3 * g, stB, fB);
      math(1);
    }
  }
  switch (w) {
    case 0: gram_store<MC, 0, 0>(acc, lane, col, gpart); break;
    case 1: gram_store<MC, 0, 1>(acc, lane, col, gpart); break;
    case 2: gram_store<MC, 1, 0>(acc, lane, col, gpart); break;
    case 3: gram_store<MC, 1, 1>(acc, lane, col, gpart); break;
    case 4: gram_store<MC, 2, 0>(acc, lane, col, gpart); break;
    case 5: gram_store<MC, 2, 1>(acc, lane, col, gpart); break;
    case 6: gram_store<MC, 3, 0>(acc, lane, col, gpart); break;
    default: gram_store<MC, 3, 1>(acc, lane, col, gpart); break;
  }
}

template <typename T>
void launch_formk_gram(Queue &q, int64_t n, WStore<T> w, int head, int col,
                       const iw_t *iwhere) {
  int gr = 0;
  if (col <= 10) {
    const int64_t nslab = (n + 127) / 128;
    gr = (int)(nslab < GRAM_BLOCKS ? nslab : GRAM_BLOCKS);
    if (col <= 5)
      hipLaunchKernelGGL((formk_gram_rows_kernel<T, 5>), dim3(gr), dim3(512), 0, q.stream, n, w.ws,
                         w.wy, w.ld, w.m, head, col, iwhere, q.d_gpart);
    else
      hipLaunchKernelGGL((formk_gram_rows_kernel<T, 10>), dim3(gr), dim3(512), 0, q.stream, n,
                         w.ws, w.wy, w.ld, w.m, head, col, iwhere, q.d_gpart);
    q.launches++;
    finalize_from(q, q.d_gpart, GRAM_BLOCKS, gr, 2 * col * col + col, 0, 0);
    return;
  }
  DISPATCH_MAXC(col, {
    const int64_t ntiles = (n + GramCfg<MC>::R - 1) / GramCfg<MC>::R;
    gr = (int)(ntiles < GRAM_BLOCKS ? ntiles : GRAM_BLOCKS);
    hipLaunchKernelGGL((formk_gram_kernel<T, MC>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws,
                       w.wy, w.ld, w.m, head, col, iwhere, q.d_gpart);
  });
  q.launches++;
  finalize_from(q, q.d_gpart, GRAM_BLOCKS, gr, 2 * col * col + col, 0, 0);
}

// The generalized Cauchy point is not stored as a vector on the main path: after the walk,
// xcp(k) is a function of row k's own x, g, bounds and iwhere (cauchy :1341, :1425-1433, :1515):
//   iwhere in {0,-1} (the row moves with d = -g and was not fixed):  x + tsum*d
//   iwhere == 1 / 2 (at its lower/upper bound, before or by this walk): that bound
//   otherwise (always fixed, or free with zero gradient): x
// Every consumer evaluates exactly the expression cauchy_finish_kernel stores (incl. the
// rounding to T), so results do not depend on whether z was materialised.
template <typename T>
__device__ __forceinline__ double xcp_free(double xk, double gk, int iw, double tsum) {
  if ((iw == 0 || iw == -1) && tsum != 0.0) return (double)(T)(xk + tsum * (-gk));
  return xk;
}
template <typename T>
__device__ __forceinline__ double xcp_row(double xk, double gk, int iw, double lk, double uk,
                                          double tsum) {
  if (tsum == 0.0) return xk;  // a walk that fixed a row has tsum >= its breakpoint > 0
  if (iw == 1) return xk == lk ? xk : lk;
  if (iw == 2) return xk == uk ? xk : uk;
  return xcp_free<T>(xk, gk, iw, tsum);
}

// =========================== cmprlb (:1548-1586) =============================
// cmprlb fused with the first matvec of subsm (:2742-2754): r_k depends only on row k, so
// W'r is accumulated in the same pass that computes r (one pass over W instead of two).  Per
// element the arithmetic is the reference's: r = -theta (z - x) - g, then + Wy(k,j) a1_j +
// Ws(k,j) a2_j for j = 1..col in that order (:1565-1583).
// NEWROW: the same pass also yields the new row/column of formk's WN1 (:1756-1793) for the
// pair just stored (logical column col-1): with y = Wy_new, s = Ws_new,
//   t1_j = sum_free y Wy_j, t2_j = sum_act s Ws_j, t3_j = sum_act s Wy_j, t4_j = sum_free Ws_j y.
// slots: [0,MC) Wy'r | [MC,2MC) Ws'r | NEWROW: [2MC,3MC) t1 | [3MC,4MC) t2 | [4MC,5MC) t3 | [5MC,6MC) t4
// r itself is NOT stored: its only consumer, subsm_update_kernel, streams the same operands
// anyway and recomputes it bit for bit (a store stream costs this HBM-bound pass more than it
// moves: +0.8 GB written = +0.45 ms at n = 1e8, profiles/scripts/cmprlb_wtv_variants.hip).
template <typename T, int MC, bool NEWROW, bool NT>
__global__ __launch_bounds__(BLOCK) void cmprlb_wtv_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ g, double tsum,
    const iw_t *__restrict__ iwhere, const T *__restrict__ ws, const T *__restrict__ wy,
    int64_t ldw, int m, int head, int col, double theta, Coef cf, int plain, const T *pr,
    const T *pd, Pend pe, double *part) {
  constexpr int NA = NEWROW ? 6 * MC : 2 * MC;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  for_rows<T, RowsPerAcc<T, MC, NA>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], gv[W], rv[W], a[MC][W], b[MC][W];
    int iw[W];
    ldx<W, NT>(g + i, gv);
    if (!plain) {
      ldx<W, NT>(x + i, xv);
      ldi<W>(iwhere + i, iw);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) iw[k] = -1;  // unconstrained: every row is free
    }
    load_cols<T, MC, W, NT>(wy, ws, pr, pd, i, col, head, m, ldw, pe, a, b);
    fix_pending<T, MC, W>(col, pe, gv, a, b);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (plain) {  // unconstrained and col > 0: r = -g (:1560-1563)
        rv[k] = -gv[k];
      } else {
        const double zk = xcp_free<T>(xv[k], gv[k], iw[k], tsum);  // only free rows are used
        double rr = -theta * (zk - xv[k]) - gv[k];
#pragma unroll
        for (int j = 0; j < MC; ++j) {
          if (j < col) rr = rr + a[j][k] * cf.a[j] + b[j][k] * cf.a[MAXM + j];
        }
        rv[k] = iw[k] <= 0 ? rr : 0.0;
      }
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += a[j][k] * rv[k];
        acc[MC + j] += b[j][k] * rv[k];
      }
    }
    if constexpr (NEWROW) {
      double yf[W], sa[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        double yn = 0.0, sn = 0.0;
#pragma unroll
        for (int j = 0; j < MC; ++j)
          if (j == col - 1) {
            yn = a[j][k];
            sn = b[j][k];
          }
        yf[k] = iw[k] <= 0 ? yn : 0.0;  // free rows
        sa[k] = iw[k] <= 0 ? 0.0 : sn;  // active rows
      }
#pragma unroll
      for (int j = 0; j < MC; ++j) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
          acc[2 * MC + j] += yf[k] * a[j][k];  // temp1 (:1764)
          acc[3 * MC + j] += sa[k] * b[j][k];  // temp2 (:1769)
          acc[4 * MC + j] += sa[k] * a[j][k];  // temp3 (:1770)
          acc[5 * MC + j] += b[j][k] * yf[k];  // temp3 of the new column (:1789)
        }
      }
    }
  });
  block_reduce_store<NA>(acc, NA, 0, 0, part, MAX_BLOCKS);
}
// The same pass for MC >= 20 with the new-row sums: 6*MC accumulators per lane do not fit the
// register file (they spill to AGPRs and the pass runs one wave per SIMD at ~4 TB/s).  Here two
// neighbouring lanes share the work on their two row groups: every lane still loads all columns
// of its own rows (r needs them), but accumulates only ONE HALF of the columns -- for its own
// rows and, through lane shuffles, for its neighbour's -- so 6*MC/2 accumulators suffice, and the
// operands stay in storage precision until they are used.  Per-element arithmetic is unchanged;
// the sums are merely grouped differently.  Rows without a neighbour (odd group count, scalar
// tail) are loaded by both lanes 0 and 1 of the first workgroup, each taking its half.
// fp64, m = 20, n = 1e8: 7.5 -> 5.4 ms (4.5 -> 6.3 TB/s).  Used for fp64 only: the fp32 build of
// it is slower than the plain kernel (operand widening + ds_bpermute shuffles).
template <typename T, int W, bool NT>
__device__ __forceinline__ void ldraw(const T *p, T (&o)[W]) {
  double t[W];
  ldx<W, NT>(p, t);
#pragma unroll
  for (int k = 0; k < W; ++k) o[k] = (T)t[k];  // exact: t came from a T
}
// fp32 operand -> fp64 at the point of use.  Opaque to the optimiser on purpose: a plain cast
// would be hoisted and shared with the earlier use, keeping all operands live as doubles.
__device__ __forceinline__ double widen_late(float v) {
  double d;
  asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(v));
  return d;
}
__device__ __forceinline__ double widen_late(double v) { return v; }
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void cmprlb_wtv_pair_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ g, double tsum,
    const iw_t *__restrict__ iwhere, const T *__restrict__ ws, const T *__restrict__ wy,
    int64_t ldw, int m, int head, int col, double theta, Coef cf, int plain, const T *pr,
    const T *pd, Pend pe, double *part) {
  constexpr int H = MC / 2, NA = 6 * H;
  constexpr int V = RowsPer<T, MC>::V;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const int lane = threadIdx.x & 63;
  const bool hi = lane & 1;  // this lane sums columns [H, MC), its neighbour [0, H)
  const int64_t dy = pe.on ? (int64_t)(((intptr_t)pr - (intptr_t)wy) / (intptr_t)sizeof(T)) : 0;
  const int64_t ds = pe.on ? (int64_t)(((intptr_t)pd - (intptr_t)ws) / (intptr_t)sizeof(T)) : 0;
  auto process = [&](int64_t i, auto wt, bool paired) {
    constexpr int W = decltype(wt)::value;
    double xv[W], gv[W], rv[W], yf[W], sa[W];
    T a[MC][W], b[MC][W];
    int iw[W];
    ldx<W, NT>(g + i, gv);
    if (!plain) {
      ldx<W, NT>(x + i, xv);
      ldi<W>(iwhere + i, iw);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) iw[k] = -1;
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, col, head, m, ldw);
      const bool pj = pe.on && j == col - 1;
      ldraw<T, W, NT>(wy + ((pj ? dy : off) + i), a[j]);
      ldraw<T, W, NT>(ws + ((pj ? ds : off) + i), b[j]);
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {  // the pending column, with the rounding of a store
      const bool pj = pe.on && j == col - 1;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const T yk = (T)pend_y<T>(gv[k], (double)a[j][k]);
        const T sk = (T)pend_s<T>((double)b[j][k], pe.stp);
        a[j][k] = pj ? yk : a[j][k];
        b[j][k] = pj ? sk : b[j][k];
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      double yn = 0.0, sn = 0.0;
      if (plain) {
        rv[k] = -gv[k];
      } else {
        const double zk = xcp_free<T>(xv[k], gv[k], iw[k], tsum);
        double rr = -theta * (zk - xv[k]) - gv[k];
#pragma unroll
        for (int j = 0; j < MC; ++j)
          if (j < col) rr = rr + (double)a[j][k] * cf.a[j] + (double)b[j][k] * cf.a[MAXM + j];
        rv[k] = iw[k] <= 0 ? rr : 0.0;
      }
#pragma unroll
      for (int j = 0; j < MC; ++j)
        if (j == col - 1) {
          yn = (double)a[j][k];
          sn = (double)b[j][k];
        }
      yf[k] = iw[k] <= 0 ? yn : 0.0;
      sa[k] = iw[k] <= 0 ? 0.0 : sn;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      double prv = 0.0, pyf = 0.0, psa = 0.0;
      if (paired) {
        prv = __shfl_xor(rv[k], 1);
        pyf = __shfl_xor(yf[k], 1);
        psa = __shfl_xor(sa[k], 1);
      }
#pragma unroll
      for (int jj = 0; jj < H; ++jj) {
        // own rows, own half of the columns
        const double aj = widen_late(hi ? a[H + jj][k] : a[jj][k]);
        const double bj = widen_late(hi ? b[H + jj][k] : b[jj][k]);
        acc[jj] += aj * rv[k];
        acc[H + jj] += bj * rv[k];
        acc[2 * H + jj] += yf[k] * aj;
        acc[3 * H + jj] += sa[k] * bj;
        acc[4 * H + jj] += sa[k] * aj;
        acc[5 * H + jj] += bj * yf[k];
        if (paired) {  // the neighbour's rows: it sends the half it does not sum itself
          const T sa_ = hi ? a[jj][k] : a[H + jj][k];
          const T sb_ = hi ? b[jj][k] : b[H + jj][k];
          const double paj = widen_late(__shfl_xor(sa_, 1));
          const double pbj = widen_late(__shfl_xor(sb_, 1));
          acc[jj] += paj * prv;
          acc[H + jj] += pbj * prv;
          acc[2 * H + jj] += pyf * paj;
          acc[3 * H + jj] += psa * pbj;
          acc[4 * H + jj] += psa * paj;
          acc[5 * H + jj] += pbj * pyf;
        }
      }
    }
  };
  const int64_t nv = n / V, nve = nv & ~(int64_t)1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; iv < nve; iv += stride)
    process(iv * V, WTag<V>{}, true);
  if (blockIdx.x == 0 && threadIdx.x < 2) {  // rows without a neighbour: both lanes, one half each
    if (nv > nve) process(nve * V, WTag<V>{}, false);
    for (int64_t rrow = nv * V; rrow < n; ++rrow) process(rrow, WTag<1>{}, false);
  }
  // lanes of equal parity hold the same slots: reduce over them, then across the 4 waves
  __shared__ double sm[4][2][NA];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane < 2) sm[w][lane][k] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * NA; e += blockDim.x) {
    const int par = e / NA, k = e % NA;
    const double sum = ((sm[0][par][k] + sm[1][par][k]) + sm[2][par][k]) + sm[3][par][k];
    const int grp = k / H, jj = k % H;
    part[(size_t)(grp * MC + par * H + jj) * MAX_BLOCKS + blockIdx.x] = sum;
  }
}

template <typename T>
void launch_cmprlb_wtv(Queue &q, int64_t n, const T *x, const T *g, double tsum,
                       const iw_t *iwhere, WStore<T> w, int head, int col, double theta,
                       const Coef &a, int plain, int newrow, const T *pr, const T *pd, Pend pe) {
  const int gr = grid_for_w(n, VecOf<T>::V, (int)sizeof(T));
  if (newrow && maxc_for(col) >= 20 && sizeof(T) == 8) {  // (fp32: the plain kernel is faster)
    if (maxc_for(col) == 20) {
      if (q.nt)
        hipLaunchKernelGGL((cmprlb_wtv_pair_kernel<T, 20, true>), dim3(gr), dim3(BLOCK), 0, q.stream, n, x,
                           g, tsum, iwhere, w.ws, w.wy, w.ld, w.m, head, col, theta, a, plain, pr, pd,
                           pe, q.d_part);
      else
        hipLaunchKernelGGL((cmprlb_wtv_pair_kernel<T, 20, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n,
                           x, g, tsum, iwhere, w.ws, w.wy, w.ld, w.m, head, col, theta, a, plain, pr,
                           pd, pe, q.d_part);
    } else {
      if (q.nt)
        hipLaunchKernelGGL((cmprlb_wtv_pair_kernel<T, 32, true>), dim3(gr), dim3(BLOCK), 0, q.stream, n, x,
                           g, tsum, iwhere, w.ws, w.wy, w.ld, w.m, head, col, theta, a, plain, pr, pd,
                           pe, q.d_part);
      else
        hipLaunchKernelGGL((cmprlb_wtv_pair_kernel<T, 32, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n,
                           x, g, tsum, iwhere, w.ws, w.wy, w.ld, w.m, head, col, theta, a, plain, pr,
                           pd, pe, q.d_part);
    }
  } else if (newrow) {
    DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((cmprlb_wtv_kernel<T, MC, true, NTV>), dim3(gr), dim3(BLOCK), 0,
                                          q.stream, n, x, g, tsum, iwhere, w.ws, w.wy, w.ld, w.m,
                                          head, col, theta, a, plain, pr, pd, pe, q.d_part));
  } else {
    DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((cmprlb_wtv_kernel<T, MC, false, NTV>), dim3(gr), dim3(BLOCK), 0,
                                          q.stream, n, x, g, tsum, iwhere, w.ws, w.wy, w.ld, w.m,
                                          head, col, theta, a, plain, pr, pd, pe, q.d_part));
  }
  q.launches++;
  launch_finalize(q, gr, (newrow ? 6 : 2) * maxc_for(col), 0, 0);
}

// formk's patches for variables that changed status (:1801-1851): signed Gram over the
// listed rows only, sign +1 for rows that entered the free set, -1 for rows that left it.
// chg[k] = local row | (left ? 0x80000000 : 0).  Output layout = the Gram's (E entries for
// `upcl` columns): P_yy (i>=j), P_ss (i>=j), P_sy (all i,j).
template <typename T>
__global__ __launch_bounds__(BLOCK) void formk_patch_kernel(const uint32_t *__restrict__ chg,
                                                            uint32_t cnt,
                                                            const T *__restrict__ ws,
                                                            const T *__restrict__ wy, int64_t ldw,
                                                            int m, int head, int upcl,
                                                            double *gpart) {
  constexpr int R = 64, RS = 2 * MAXM + 1;
  __shared__ double tile[R * RS];
  __shared__ double sgn[R];
  const int tri = upcl * (upcl + 1) / 2;
  const int E = 2 * upcl * upcl + upcl;
  constexpr int NE = (2 * MAXM * MAXM + MAXM + BLOCK - 1) / BLOCK;
  int ca[NE], cb[NE];
  double acc[NE];
#pragma unroll
  for (int s = 0; s < NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    acc[s] = 0.0;
    ca[s] = cb[s] = 0;
    if (e < E) {
      if (e < 2 * tri) {
        const int ee = e < tri ? e : e - tri;
        int i = (int)((sqrt(8.0 * ee + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > ee) --i;
        while ((i + 1) * (i + 2) / 2 <= ee) ++i;
        const int j = ee - i * (i + 1) / 2;
        ca[s] = (e < tri ? 0 : upcl) + i;
        cb[s] = (e < tri ? 0 : upcl) + j;
      } else {
        const int ee = e - 2 * tri;
        ca[s] = upcl + ee / upcl;  // Ws_i
        cb[s] = ee % upcl;         // Wy_j
      }
    }
  }
  const uint32_t ntile = (cnt + R - 1) / R;
  for (uint32_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    __syncthreads();
    for (int qd = threadIdx.x; qd < 2 * upcl * R; qd += BLOCK) {
      const int c = qd / R, rr = qd % R;
      const uint32_t k = t * R + rr;
      double v = 0.0;
      if (k < cnt) {
        const int64_t row = chg[k] & 0x7FFFFFFFu;
        const int jj = c < upcl ? c : c - upcl;
        const int64_t off = (int64_t)((head - 1 + jj) % m) * ldw + row;
        v = c < upcl ? (double)wy[off] : (double)ws[off];
      }
      tile[rr * RS + c] = v;
    }
    for (int rr = threadIdx.x; rr < R; rr += BLOCK) {
      const uint32_t k = t * R + rr;
      sgn[rr] = k < cnt ? ((chg[k] & 0x80000000u) ? -1.0 : 1.0) : 0.0;
    }
    __syncthreads();
    for (int rr = 0; rr < R; ++rr) {
      const double sg = sgn[rr];
#pragma unroll
      for (int s = 0; s < NE; ++s) acc[s] += sg * tile[rr * RS + ca[s]] * tile[rr * RS + cb[s]];
    }
  }
#pragma unroll
  for (int s = 0; s < NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    if (e < E) gpart[(size_t)e * GRAM_BLOCKS + blockIdx.x] = acc[s];
  }
}
template <typename T>
void launch_formk_patch(Queue &q, const uint32_t *chg, uint32_t cnt, WStore<T> w, int head, int upcl) {
  int gr = (int)((cnt + 63) / 64);
  if (gr < 1) gr = 1;
  if (gr > 256) gr = 256;
  hipLaunchKernelGGL(formk_patch_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, chg, cnt, w.ws, w.wy,
                     w.ld, w.m, head, upcl, q.d_gpart);
  q.launches++;
  finalize_from(q, q.d_gpart, GRAM_BLOCKS, gr, 2 * upcl * upcl + upcl, 0, 0);
}

// =========================== subsm (:2676-2885) ==============================
// Newton direction of one free row (cmprlb :1560-1583 then subsm :2770-2780): the reduced
// gradient r is recomputed here exactly as cmprlb_wtv_kernel computed it for W'r.
template <int MC>
__device__ __forceinline__ double subsm_dir(double xk, double zk, double gk, const double (&a)[MC],
                                            const double (&b)[MC], int col, double theta,
                                            double rtheta, const Coef &cf, int plain,
                                            const Coef &wv) {
  double dk;
  if (plain) {
    dk = -gk;
  } else {
    dk = -theta * (zk - xk) - gk;
#pragma unroll
    for (int j = 0; j < MC; ++j)
      if (j < col) dk = dk + a[j] * cf.a[j] + b[j] * cf.a[MAXM + j];
  }
#pragma unroll
  for (int j = 0; j < MC; ++j)
    if (j < col) dk = dk + a[j] * wv.a[j] / theta + b[j] * wv.a[MAXM + j];
  return rtheta * dk;  // dscal (:2780)
}

// One pass: Newton direction, projected step (:2789-2816), dd_p (:2824-2827) and -- because the
// projected point is final unless the rare backtracking branch (:2830-2879) is taken -- what
// mainlb :720-722 and the first call of lnsrlb (:2196-2236) do next: d = z - x, t = x, r = g,
// dtd = d'd, the stpmx ratios; g'd is dd_p itself.  The Cauchy point is evaluated per row
// (xcp_row), the subspace minimiser written to `zout`; neither xp (:2787) nor the direction is
// stored (the backtracking branch regenerates both: cauchy_finish_kernel, subsm_dir_kernel).
// res: sum [0] = #bound hits (iword), [1] = dd_p (= g'd), [2] = dtd ; min [3] = stpmx
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void subsm_update_kernel(
    int64_t n, double tsum, T *__restrict__ zout, T *r,
    const T *__restrict__ l, const T *__restrict__ u, const int32_t *__restrict__ nbd,
    const iw_t *__restrict__ iwhere, const T *xx, const T *__restrict__ gg,
    const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw, int m, int head, int col,
    double theta, Coef cf, int plain, Coef wv, T *dvec, T *__restrict__ tvec,
    T *xout, int do_stpmx, Pend pe, T *cwy, T *cws, double *part) {
  double acc[4] = {0.0, 0.0, 0.0, 1.0e10};
  const double rtheta = 1.0 / theta;
  for_rows<T, RowsPer<T, MC>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double zv[W], lv[W], uv[W], xv[W], gv[W], a[MC][W], b[MC][W];
    int nb[W], iw[W];
    ldx<W, NT>(l + i, lv);
    ldx<W, NT>(u + i, uv);
    ldx<W, NT>(xx + i, xv);
    ldx<W, NT>(gg + i, gv);
    ldi<W>(nbd + i, nb);
    if (!plain) {
      ldi<W>(iwhere + i, iw);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) iw[k] = -1;  // unconstrained: every row is free
    }
    // a pending pair is read from (r, d) -- which this pass overwrites further down -- and
    // committed to its W slot (cwy, cws) here
    load_cols<T, MC, W, NT>(wy, ws, r, dvec, i, col, head, m, ldw, pe, a, b);
    fix_pending<T, MC, W>(col, pe, gv, a, b);
    if (pe.on) {
      double yn[W], sn[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        yn[k] = 0.0, sn[k] = 0.0;
#pragma unroll
        for (int j = 0; j < MC; ++j)
          if (j == col - 1) {
            yn[k] = a[j][k];
            sn[k] = b[j][k];
          }
      }
      if (NT) {
        stnt<W>(cwy + i, yn);
        stnt<W>(cws + i, sn);
      } else {
        st<W>(cwy + i, yn);
        st<W>(cws + i, sn);
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) zv[k] = xcp_row<T>(xv[k], gv[k], iw[k], lv[k], uv[k], tsum);
    double dv[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (iw[k] <= 0) {
        double ak[MC], bk[MC];
#pragma unroll
        for (int j = 0; j < MC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
        const double dk = subsm_dir<MC>(xv[k], zv[k], gv[k], ak, bk, col, theta, rtheta, cf, plain, wv);
        const double xk = zv[k];
        if (nb[k] != 0) {
          if (nb[k] == 1) {
            zv[k] = fmax(lv[k], xk + dk);
            if (zv[k] == lv[k]) acc[0] += 1.0;
          } else if (nb[k] == 2) {
            const double t1 = fmax(lv[k], xk + dk);
            zv[k] = fmin(uv[k], t1);
            if (zv[k] == lv[k] || zv[k] == uv[k]) acc[0] += 1.0;
          } else if (nb[k] == 3) {
            zv[k] = fmin(uv[k], xk + dk);
            if (zv[k] == uv[k]) acc[0] += 1.0;
          }
        } else {
          zv[k] = xk + dk;
        }
      }
      dv[k] = zv[k] - xv[k];            // mainlb :720-722
      acc[1] = acc[1] + dv[k] * gv[k];  // dd_p (:2824-2827) == g'd (:2244)
      acc[2] = acc[2] + dv[k] * dv[k];  // dtd (:2196)
      if (do_stpmx && nb[k] != 0) {     // :2206-2225
        const double a1 = dv[k];
        if (a1 < 0.0 && nb[k] <= 2) {
          const double a2 = lv[k] - xv[k];
          acc[3] = fmin(acc[3], a2 >= 0.0 ? 0.0 : a2 / a1);
        } else if (a1 > 0.0 && nb[k] >= 2) {
          const double a2 = uv[k] - xv[k];
          acc[3] = fmin(acc[3], a2 <= 0.0 ? 0.0 : a2 / a1);
        }
      }
    }
    // (nontemporal stores with the nontemporal loads: large problems, nothing here is re-read
    //  by this pass; the next readers stream it from HBM anyway)
    if (NT) {
      stnt<W>(zout + i, zv);
      stnt<W>(dvec + i, dv);
      stnt<W>(tvec + i, xv);
      stnt<W>(r + i, gv);
      if (xout) stnt<W>(xout + i, zv);
    } else {
      st<W>(zout + i, zv);
      st<W>(dvec + i, dv);
      st<W>(tvec + i, xv);  // t = x (:2235)
      st<W>(r + i, gv);     // r = g (:2236)
      // first trial point of the line search when its step is known to be 1: x = z (:2265);
      // xout aliases xx (each row is read above before it is written here)
      if (xout) st<W>(xout + i, zv);
    }
  });
  block_reduce_store<4>(acc, 3, 1, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_subsm_update(Queue &q, int64_t n, double tsum, T *zout, T *r, const T *l, const T *u,
                         const int32_t *nbd, const iw_t *iwhere, const T *xx, const T *gg,
                         WStore<T> w, int head, int col, double theta, const Coef &cf, int plain,
                         const Coef &wv, T *dvec, T *tvec, T *xout, int do_stpmx, Pend pe) {
  const int gr = grid_for_w(n, VecOf<T>::V, (int)sizeof(T));
  const int64_t slot = (int64_t)((head - 1 + col - 1) % w.m) * w.ld;  // physical column of col-1
  DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((subsm_update_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                        q.stream, n, tsum, zout, r, l, u, nbd, iwhere, xx, gg, w.ws,
                                        w.wy, w.ld, w.m, head, col, theta, cf, plain, wv, dvec, tvec,
                                        xout, do_stpmx, pe, w.wy + slot, w.ws + slot, q.d_part));
  q.launches++;
  launch_finalize(q, gr, 3, 1, 0);
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
template <typename T>
void launch_pair_commit(Queue &q, int64_t n, const T *g, const T *r, const T *d, Pend pe,
                        WStore<T> w, int head, int col) {
  const int gr = grid_for(n, VecOf<T>::V);
  const int64_t slot = (int64_t)((head - 1 + col - 1) % w.m) * w.ld;
  hipLaunchKernelGGL(pair_commit_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, g, r, d, pe.stp,
                     w.wy + slot, w.ws + slot);
  q.launches++;
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
  q.launches++;
}

// The Newton direction as a vector (free rows; 0 elsewhere), for the backtracking branch only.
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void subsm_dir_kernel(
    int64_t n, const T *__restrict__ xcp, const iw_t *__restrict__ iwhere,
    const T *__restrict__ xx, const T *__restrict__ gg, const T *__restrict__ ws,
    const T *__restrict__ wy, int64_t ldw, int m, int head, int col, double theta, Coef cf,
    int plain, Coef wv, T *__restrict__ ndir) {
  const double rtheta = 1.0 / theta;
  for_rows<T, RowsPer<T, MC>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double zv[W], xv[W], gv[W], a[MC][W], b[MC][W], out[W];
    int iw[W];
    ldx<W, NT>(xcp + i, zv);
    ldx<W, NT>(xx + i, xv);
    ldx<W, NT>(gg + i, gv);
    if (!plain) {
      ldi<W>(iwhere + i, iw);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) iw[k] = -1;
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, col, head, m, ldw) + i;
      ld_col<T, W, NT>(j < col, wy + off, a[j]);
      ld_col<T, W, NT>(j < col, ws + off, b[j]);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      out[k] = 0.0;
      if (iw[k] <= 0) {
        double ak[MC], bk[MC];
#pragma unroll
        for (int j = 0; j < MC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
        out[k] = subsm_dir<MC>(xv[k], zv[k], gv[k], ak, bk, col, theta, rtheta, cf, plain, wv);
      }
    }
    st<W>(ndir + i, out);
  });
}
template <typename T>
void launch_subsm_dir(Queue &q, int64_t n, const T *xcp, const iw_t *iwhere, const T *xx,
                      const T *gg, WStore<T> w, int head, int col, double theta, const Coef &cf,
                      int plain, const Coef &wv, T *ndir) {
  const int gr = grid_for_w(n, VecOf<T>::V, (int)sizeof(T));
  DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((subsm_dir_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                        q.stream, n, xcp, iwhere, xx, gg, w.ws, w.wy, w.ld, w.m, head,
                                        col, theta, cf, plain, wv, ndir));
  q.launches++;
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
                     r, l, u, nbd, iwhere, 0, 0.0, q.d_part);
  q.launches++;
  launch_finalize(q, gr, 0, 1, 0);
}
template <typename T>
void launch_subsm_argalpha(Queue &q, int64_t n, int64_t row0, const T *xp, const T *r, const T *l,
                           const T *u, const int32_t *nbd, const iw_t *iwhere, double alpha) {
  const int gr = grid_for(n, 1);
  hipLaunchKernelGGL(subsm_alpha_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, xp, r, l,
                     u, nbd, iwhere, 1, alpha, q.d_part);
  q.launches++;
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
  q.launches++;
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
    st<W>(t + i, xv);
    st<W>(r + i, gv);
  });
  block_reduce_store<3>(acc, 2, 1, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_lnsrlb_begin(Queue &q, int64_t n, const T *z, const T *x, const T *g, const T *l,
                         const T *u, const int32_t *nbd, T *d, T *t, T *r, int do_stpmx) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(lnsrlb_begin_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, z, x, g, l, u,
                     nbd, d, t, r, do_stpmx, q.d_part);
  q.launches++;
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
  q.launches++;
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
                     d, q.d_part);
  q.launches++;
  launch_finalize(q, gr, 1, 0, 1);
}

// =========================== mainlb :812-824 + matupd (:2291-2346) ===========
// ncol_old = col - 1 older pairs (logical order from head); new pair goes to
// physical column itail (1-based).
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void update_pairs_kernel(
    int64_t n, const T *__restrict__ g, const T *__restrict__ r, const T *__restrict__ d,
    double stp, T *ws, T *wy, int64_t ldw, int m, int head, int nold, int itail, double *part) {
  constexpr int NA = 2 * MC + 1;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const int64_t offn = (int64_t)(itail - 1) * ldw;
  for_rows<T, RowsPer<T, MC>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double gv[W], rv[W], dv[W], a[MC][W], b[MC][W];
    ldx<W, NT>(g + i, gv);
    ldx<W, NT>(r + i, rv);
    ldx<W, NT>(d + i, dv);
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      // nold may be 0: then logical column 0 is the NEW column; read d's own slot instead
      const int64_t off = (nold > 0 ? col_off(j, nold, head, m, ldw) : offn) + i;
      ld_col<T, W, NT>(j < nold, wy + off, a[j]);
      ld_col<T, W, NT>(j < nold, ws + off, b[j]);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      rv[k] = gv[k] - rv[k];                   // y = g - g_old (:813-815)
      acc[2 * MC] = acc[2 * MC] + rv[k] * rv[k];  // rr (:816)
      if (stp != 1.0) dv[k] = stp * dv[k];     // dscal (:822)
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += dv[k] * a[j][k];        // Sy(col,j) = d . Wy(:,j) (:2335)
        acc[MC + j] += b[j][k] * dv[k];   // Ss(j,col) = Ws(:,j) . d (:2336)
      }
    }
    st<W>(ws + offn + i, dv);  // :2313
    st<W>(wy + offn + i, rv);  // :2314
  });
  // slots [0..MC) d'Wy_j, [MC..2MC) Ws_j'd, [2MC] y'y
  block_reduce_store<NA>(acc, NA, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_update_pairs(Queue &q, int64_t n, const T *g, const T *r, const T *d, double stp,
                         WStore<T> w, int head, int col, int itail) {
  const int gr = grid_for_w(n, VecOf<T>::V, (int)sizeof(T));
  const int nold = col - 1;
  DISPATCH_MAXC_NT(nold, q.nt, hipLaunchKernelGGL((update_pairs_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                         q.stream, n, g, r, d, stp, w.ws, w.wy, w.ld, w.m, head,
                                         nold, itail, q.d_part));
  q.launches++;
  launch_finalize(q, gr, 2 * maxc_for(nold) + 1, 0, 0);
}

// =========================== matupd + cauchy scan, fused ======================
// On a NEW_X re-entry the reference runs matupd (:842) and, at the top of the next loop
// trip, the n-loop of cauchy (:1270-1330).  Both stream every stored column of W; fused,
// the old columns are read ONCE for s'Wy_j, Ws_j's (matupd) and for p = W'd (cauchy), and
// the new pair (s, y) is used from registers.  Per element the arithmetic is exactly that
// of update_pairs_kernel and cauchy_scan_kernel.
// slots: [0,MC) s'Wy_j | [MC,2MC) Ws_j's | [2MC] y'y | [2MC+1,3MC+1) Wy_j'd | [3MC+1] y'd |
//        [3MC+2,4MC+2) Ws_j'd | [4MC+2] s'd | f1, nbreak, nunb, nunbnz | [4MC+7] g'd (unscaled d)
//        | [4MC+8] #rows whose iwhere changed | min [4MC+9] bkmin | max [4MC+10] |proj g|
// The same pass serves as the line search's evaluation at a trial point (g'd, |proj g|): run
// speculatively there (store_pair = 0; it writes nothing but -- with store_iw -- the few iwhere
// entries that changed), its sums ARE the matupd + cauchy-scan results if the trial is accepted.
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void update_scan_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const int32_t *__restrict__ nbd, const T *__restrict__ g, const T *__restrict__ r,
    const T *__restrict__ d, double stp, iw_t *iwhere, T *tbrk, T *ws, T *wy, int64_t ldw,
    int m, int head, int nold, int itail, int store_pair, int store_iw, double *part) {
  constexpr int NA = 4 * MC + 11;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  acc[4 * MC + 9] = LB_INF;
  const int64_t offn = (int64_t)(itail - 1) * ldw;
  for_rows<T, RowsPerAcc<T, MC, NA>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], rv[W], dv[W], tb[W], ng[W], a[MC][W], b[MC][W];
    int nb[W], iw[W];
    ldx<W, NT>(x + i, xv);
    ldx<W, NT>(l + i, lv);
    ldx<W, NT>(u + i, uv);
    ldx<W, NT>(g + i, gv);
    ldx<W, NT>(r + i, rv);
    ldx<W, NT>(d + i, dv);
    ldi<W>(nbd + i, nb);
    ldi<W>(iwhere + i, iw);
    bool iw_changed = false;
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = (nold > 0 ? col_off(j, nold, head, m, ldw) : offn) + i;
      ld_col<T, W, NT>(j < nold, wy + off, a[j]);
      ld_col<T, W, NT>(j < nold, ws + off, b[j]);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      // ---- the line search's own sums at this trial point: g'd (:2244), |proj g| (:781) ----
      acc[4 * MC + 7] = acc[4 * MC + 7] + gv[k] * dv[k];
      acc[4 * MC + 10] = fmax(acc[4 * MC + 10], proj_g(xv[k], lv[k], uv[k], nb[k], gv[k]));
      rv[k] = gv[k] - rv[k];                              // y (:813-815)
      acc[2 * MC] = acc[2 * MC] + rv[k] * rv[k];          // rr (:816)
      if (stp != 1.0) dv[k] = stp * dv[k];                // s (:822)
      // ---- cauchy n-loop (:1270-1330) ----
      const double neggi = -gv[k];
      double tl = 0.0, tu = 0.0;
      if (iw[k] != 3 && iw[k] != -1) {
        if (nb[k] <= 2) tl = xv[k] - lv[k];
        if (nb[k] >= 2) tu = uv[k] - xv[k];
        const bool xlower = nb[k] <= 2 && tl <= 0.0;
        const bool xupper = nb[k] >= 2 && tu <= 0.0;
        const int iw_old = iw[k];
        iw[k] = 0;
        if (xlower) {
          if (neggi <= 0.0) iw[k] = 1;
        } else if (xupper) {
          if (neggi >= 0.0) iw[k] = 2;
        } else {
          if (fabs(neggi) <= 0.0) iw[k] = -3;
        }
        iw_changed = iw_changed || iw[k] != iw_old;
        if (iw[k] != iw_old) acc[4 * MC + 8] += 1.0;
      }
      if (iw[k] != 0 && iw[k] != -1) {
        tb[k] = -1.0;
        ng[k] = 0.0;
      } else {
        ng[k] = neggi;
        acc[4 * MC + 3] = acc[4 * MC + 3] - neggi * neggi;
        if (nb[k] <= 2 && nb[k] != 0 && neggi < 0.0) {
          tb[k] = tl / (-neggi);
          acc[4 * MC + 4] += 1.0;
          acc[4 * MC + 9] = fmin(acc[4 * MC + 9], tb[k]);
        } else if (nb[k] >= 2 && neggi > 0.0) {
          tb[k] = tu / neggi;
          acc[4 * MC + 4] += 1.0;
          acc[4 * MC + 9] = fmin(acc[4 * MC + 9], tb[k]);
        } else {
          tb[k] = LB_INF;
          acc[4 * MC + 5] += 1.0;
          if (fabs(neggi) > 0.0) acc[4 * MC + 6] += 1.0;
        }
      }
      acc[3 * MC + 1] += rv[k] * ng[k];  // new Wy column . d
      acc[4 * MC + 2] += dv[k] * ng[k];  // new Ws column . d
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += dv[k] * a[j][k];               // Sy(col,j) (:2335)
        acc[MC + j] += b[j][k] * dv[k];          // Ss(j,col) (:2336)
        acc[2 * MC + 1 + j] += a[j][k] * ng[k];  // p_j        (:1301)
        acc[3 * MC + 2 + j] += b[j][k] * ng[k];  // p_{col+j}  (:1302)
      }
    }
    if (store_pair) {  // else the pair stays pending (see Pend)
      st<W>(ws + offn + i, dv);
      st<W>(wy + offn + i, rv);
    }
    // iwhere settles after the first iterations: store only from waves that changed a row
    if (store_iw && __ballot(iw_changed) != 0ull) sti<W>(iwhere + i, iw);
    if (tbrk) st<W>(tbrk + i, tb);  // nullptr: the walk recomputes the times it needs
  });
  block_reduce_store<NA>(acc, 4 * MC + 9, 1, 1, part, MAX_BLOCKS);
}
template <typename T>
void launch_update_scan(Queue &q, int64_t n, const T *x, const T *l, const T *u, const int32_t *nbd,
                        const T *g, const T *r, const T *d, double stp, iw_t *iwhere, T *tbrk,
                        WStore<T> w, int head, int col, int itail, int store_pair, int store_iw) {
  const int gr = grid_for_w(n, VecOf<T>::V, (int)sizeof(T));
  const int nold = col - 1;
  DISPATCH_MAXC_NT(nold, q.nt, hipLaunchKernelGGL((update_scan_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                         q.stream, n, x, l, u, nbd, g, r, d, stp, iwhere, tbrk, w.ws,
                                         w.wy, w.ld, w.m, head, nold, itail, store_pair, store_iw,
                                         q.d_part));
  q.launches++;
  launch_finalize(q, gr, 4 * maxc_for(nold) + 9, 1, 1);
}

// =========================== built-in objectives =============================
template <typename T>
__global__ __launch_bounds__(BLOCK) void obj_quadratic_kernel(int64_t n, int64_t row0,
                                                              const T *__restrict__ x, T *g,
                                                              int nt, double *part) {
  double acc[1] = {0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], gv[W];
    ld<W>(x + i, xv);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const int64_t gi = row0 + i + k + 1;
      const double a = 1.0 + 99.0 * (double)((7919 * gi) % 10007) / 10006.0;
      const double c = -2.0 + 4.0 * (double)((104729 * gi) % 100003) / 100002.0;
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
                     q.d_part);
  q.launches++;
  launch_finalize(q, gr, 1, 0, 0);
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
                     x, g, xl, xr, q.nt ? 1 : 0, q.d_part);
  q.launches++;
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
  q.launches++;
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T)                                                                             \
  template void launch_active<T>(Queue &, int64_t, T *, const T *, const T *, const int32_t *,     \
      iw_t *, int8_t *);                                                                           \
  template void launch_errclb<T>(Queue &, int64_t, int64_t, const T *, const T *,                  \
      const int32_t *);                                                                            \
  template void launch_projgr<T>(Queue &, int64_t, const T *, const T *, const T *,                \
      const int32_t *, const T *);                                                                 \
  template void launch_wtv<T>(Queue &, int64_t, WStore<T>, int, int, const T *);                   \
  template void launch_wtv_nofinalize<T>(Queue &, int64_t, WStore<T>, int, int, const T *);        \
  template void launch_cauchy_scan<T>(Queue &, int64_t, const T *, const T *, const T *,           \
      const int32_t *, const T *, iw_t *, T *, WStore<T>, int, int);                               \
  template void launch_cauchy_window<T>(Queue &, int64_t, int64_t, const T *, double, int64_t,     \
      double, uint64_t *, uint32_t *, uint32_t, uint32_t *);                                       \
  template void launch_cauchy_allkeys<T>(Queue &, int64_t, int64_t, const T *, double, int64_t,    \
      uint64_t *, uint32_t *);                                                                     \
  template void launch_cauchy_gather<T>(Queue &, const uint32_t *, const uint64_t *, uint32_t,     \
      int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, const T *,         \
      const T *, Pend, double *);                                                                  \
  template void launch_cauchy_gather_dyn<T>(Queue &, const uint32_t *, const uint64_t *,           \
      const uint32_t *, uint32_t, int64_t, const T *, const T *, const T *, const T *,             \
      WStore<T>, int, int, const T *, const T *, Pend, double *);                                  \
  template void launch_cauchy_window_fly<T>(Queue &, int64_t, int64_t, const T *, const T *,       \
      const T *, const int32_t *, const T *, const iw_t *, double, int64_t, double, uint64_t *,    \
      uint32_t *, uint32_t, uint32_t *);                                                           \
  template void launch_iwhere_update<T>(Queue &, int64_t, const T *, const T *, const T *,         \
      const int32_t *, const T *, iw_t *);                                                         \
  template void launch_xcp_fill<T>(Queue &, int64_t, const T *, const T *, const T *,              \
      const T *, const iw_t *, double, T *);                                                       \
  template void launch_pgcp_gather<T>(Queue &, const uint32_t *, const uint64_t *, int64_t,        \
      int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, double,            \
      const T *, const T *, Pend, double *, double *, double *, double *, double *);               \
  template void launch_tbrk_fill<T>(Queue &, int64_t, const T *, const T *, const T *,             \
      const int32_t *, const T *, const iw_t *, T *);                                              \
  template void launch_cauchy_finish<T>(Queue &, int64_t, int64_t, const T *, const T *,           \
      const T *, const T *, const T *, iw_t *, T *, double, double, int64_t, int);                 \
  template void launch_formk_gram<T>(Queue &, int64_t, WStore<T>, int, int, const iw_t *);         \
  template void launch_cmprlb_wtv<T>(Queue &, int64_t, const T *, const T *, double,               \
      const iw_t *, WStore<T>, int, int, double, const Coef &, int, int, const T *, const T *,     \
      Pend);                                                                                       \
  template void launch_formk_patch<T>(Queue &, const uint32_t *, uint32_t, WStore<T>, int,         \
      int);                                                                                        \
  template void launch_subsm_update<T>(Queue &, int64_t, double, T *, T *, const T *,              \
      const T *, const int32_t *, const iw_t *, const T *, const T *, WStore<T>, int, int,         \
      double, const Coef &, int, const Coef &, T *, T *, T *, int, Pend);                          \
  template void launch_subsm_dir<T>(Queue &, int64_t, const T *, const iw_t *, const T *,          \
      const T *, WStore<T>, int, int, double, const Coef &, int, const Coef &, T *);               \
  template void launch_subsm_alpha<T>(Queue &, int64_t, const T *, const T *, const T *,           \
      const T *, const int32_t *, const iw_t *);                                                   \
  template void launch_subsm_argalpha<T>(Queue &, int64_t, int64_t, const T *, const T *,          \
      const T *, const T *, const int32_t *, const iw_t *, double);                                \
  template void launch_subsm_backtrack<T>(Queue &, int64_t, int64_t, T *, const T *, T *,          \
      const T *, const T *, const iw_t *, double, int64_t);                                        \
  template void launch_lnsrlb_begin<T>(Queue &, int64_t, const T *, const T *, const T *,          \
      const T *, const T *, const int32_t *, T *, T *, T *, int);                                  \
  template void launch_lnsrlb_step<T>(Queue &, int64_t, T *, const T *, const T *, const T *,      \
      double);                                                                                     \
  template void launch_lnsrlb_eval<T>(Queue &, int64_t, const T *, const T *, const T *,           \
      const int32_t *, const T *, const T *);                                                      \
  template void launch_update_pairs<T>(Queue &, int64_t, const T *, const T *, const T *,          \
      double, WStore<T>, int, int, int);                                                           \
  template void launch_update_scan<T>(Queue &, int64_t, const T *, const T *, const T *,           \
      const int32_t *, const T *, const T *, const T *, double, iw_t *, T *, WStore<T>, int,       \
      int, int, int, int);                                                                         \
  template void launch_pair_commit<T>(Queue &, int64_t, const T *, const T *, const T *, Pend,     \
      WStore<T>, int, int);                                                                        \
  template void launch_obj_quadratic<T>(Queue &, int64_t, int64_t, const T *, T *);                \
  template void launch_obj_rosenbrock<T>(Queue &, int64_t, int64_t, int64_t, const T *, T *,       \
      double, double);                                                                             \
  template void launch_halo_pack<T>(Queue &, int64_t, const T *, double *);
INSTANTIATE(double)
INSTANTIATE(float)

}  // namespace lbk
