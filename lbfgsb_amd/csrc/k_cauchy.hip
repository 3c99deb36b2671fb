// k_cauchy.hip -- cauchy: the n-loop (scan), breakpoint windows, record gathers, finish / fix
// (part of the gfx950 kernel set; kernels_common.hpp has the overview; the ordering of breakpoints is in
//  k_sort.hip, the opt-in parallel search in k_pgcp.hip, freev in k_freev.hip)
#include "kernels_common.hpp"

namespace lbk {

// =========================== cauchy scan (:1270-1330) ========================
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void cauchy_scan_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const int32_t *__restrict__ nbd, const T *__restrict__ g, iw_t *iwhere, T *tbrk,
    const T *__restrict__ ws, const T *__restrict__ wy, const T *__restrict__ zero, int64_t ldw,
    int m, int head, int col, double *part) {
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
        ld_col<T, W, NT>(j < col, wy + off, zero, a[j]);
        ld_col<T, W, NT>(j < col, ws + off, zero, b[j]);
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
  const int gr = grid_for_w(q, n, VecOf<T>::V);
  if (col == 0) {
    hipLaunchKernelGGL((cauchy_scan_kernel<T, 0, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n, x, l, u,
                       nbd, g, iwhere, tbrk, w.ws, w.wy, w.zero, w.ld, w.m, head, col, q.part());
  } else {
    DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((cauchy_scan_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                          q.stream, n, x, l, u, nbd, g, iwhere, tbrk, w.ws, w.wy,
                                          w.zero, w.ld, w.m, head, col, q.part()));
  }
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 2 * (col == 0 ? 0 : maxc_for(col)) + 4, 1, 0);
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
    // ONE atomic per wave and trip (positions lane-major inside the wave's slice): a first iteration selects
    // nearly every row, and an atomic per wave AND element -- 1.5e6 of them on one address at n = 1e8 -- made
    // this kernel 17.7 ms where its 2 GB of traffic take well under one.  The order of the output does not
    // matter (it is sorted, ties by row number, before anything reads it).
    const uint32_t c = (uint32_t)__popc(bits);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    const uint32_t total = __shfl(incl, 63);
    uint32_t base = 0;
    if (lane == 63) base = atomicAdd(count, total);
    base = __shfl(base, 63);
    uint32_t pos = base + incl - c;
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      if ((bits >> e) & 1u) {
        if (pos < cap) {
          keys[pos] = key_of(tv[e]);
          idx[pos] = (uint32_t)ri[e];
        }
        ++pos;
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
  LB_LAUNCHED(q);
}


// The window compaction without a stored tbrk: breakpoint times are recomputed per row
// (read-only pass over x, l, u, nbd, g, iwhere; the iteration's update pass then writes no
// n-vector at all, see update_scan_kernel).
template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_window_fly_kernel(
    int64_t n, int64_t row0, const T *__restrict__ x, const T *__restrict__ l,
    const T *__restrict__ u, const nb_t *__restrict__ nbd, const T *__restrict__ g,
    const iw_t *__restrict__ iwhere, double lo_t, int64_t lo_i, double hi_t, uint64_t *keys,
    uint32_t *idx, uint32_t cap, uint32_t *count, int ub) {
  const int lane = threadIdx.x & 63;
  __shared__ T dict[16];
  dict_fill<T>(dict, l, u, ub);
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], tv[W];
    int nb[W], iw[W];
    ldx<W, true>(x + i, xv);
    ldx<W, true>((ub & 1) ? l : l + i, lv);  // (uniform bounds: constant buffers, see UpdScanCtx)
    ldx<W, true>((ub & 2) ? u : u + i, uv);
    ldx<W, true>(g + i, gv);
    ldi<W>((ub & 4) ? nbd : nbd + i, nb);
    ldi<W>(iwhere + i, iw);
    dict_apply<T, W>(dict, ub, nb, lv, uv);
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      tv[k] = brk_time<T>(xv[k], lv[k], uv[k], nb[k], gv[k], iw[k]);
      const bool pred =
          tv[k] >= 0.0 && tv[k] <= hi_t && after_cursor(tv[k], row0 + i + k, lo_t, lo_i);
      bits |= pred ? (1u << k) : 0u;
    }
    if (__ballot(bits != 0) == 0ull) return;
    if (__ballot(true) == ~0ull) {
      // a full wave: ONE atomic for all its candidates of this trip (see cauchy_window_kernel)
      const uint32_t c = (uint32_t)__popc(bits);
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
      }
      const uint32_t total = __shfl(incl, 63);
      uint32_t base = 0;
      if (lane == 63) base = atomicAdd(count, total);
      base = __shfl(base, 63);
      uint32_t pos = base + incl - c;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if ((bits >> k) & 1u) {
          if (pos < cap) {
            keys[pos] = key_of(tv[k]);
            idx[pos] = (uint32_t)(i + k);
          }
          ++pos;
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {  // (the ragged end of the rows: some lanes are not here)
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
                              const nb_t *nbd, const T *g, const iw_t *iwhere, double lo_t,
                              int64_t lo_i, double hi_t, uint64_t *keys, uint32_t *idx, uint32_t cap,
                              uint32_t *d_count, int ub) {
  (void)hipMemsetAsync(d_count, 0, sizeof(uint32_t), q.stream);
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(cauchy_window_fly_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, row0, x, l,
                     u, nbd, g, iwhere, lo_t, lo_i, hi_t, keys, idx, cap, d_count, ub);
  LB_LAUNCHED(q);
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
  LB_LAUNCHED(q);
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
  LB_LAUNCHED(q);
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
  LB_LAUNCHED(q);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_gather_kernel(
    const uint32_t *__restrict__ idx, const uint64_t *__restrict__ keys, uint32_t cnt,
    int64_t row0, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const T *__restrict__ g, const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw,
    int m, int head, int col, const T *pr, const T *pd, Pend pe, double *rec, const uint64_t *__restrict__ lmask) {
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
                                      : (double)wy[(int64_t)((head - 1 + (f - 4)) % m) * ldw + wrow(lmask, i)];
    } else {
      v = (pe.on && f - 4 - col == col - 1)
              ? pend_sx<T>((double)pd[i], (double)x[i], pe)
              : (double)ws[(int64_t)((head - 1 + (f - 4 - col)) % m) * ldw + wrow(lmask, i)];
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
    const T *pr, const T *pd, Pend pe, double *msg, const uint64_t *__restrict__ lmask) {
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
                                      : (double)wy[(int64_t)((head - 1 + (f - 4)) % m) * ldw + wrow(lmask, i)];
    } else {
      v = (pe.on && f - 4 - col == col - 1)
              ? pend_sx<T>((double)pd[i], (double)x[i], pe)
              : (double)ws[(int64_t)((head - 1 + (f - 4 - col)) % m) * ldw + wrow(lmask, i)];
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
                     d_count, cap, row0, x, l, u, g, w.ws, w.wy, w.ld, w.m, head, col, pr, pd, pe, msg, w.lmask);
  LB_LAUNCHED(q);
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
                     row0, x, l, u, g, w.ws, w.wy, w.ld, w.m, head, col, pr, pd, pe, rec, w.lmask);
  LB_LAUNCHED(q);
}

// COUNT: also return the number of rows fixed (closed-form GCP, where no walk counted them)
template <typename T, bool COUNT>
__global__ __launch_bounds__(BLOCK) void cauchy_finish_kernel(
    int64_t n, int64_t row0, const T *__restrict__ x, const T *__restrict__ l,
    const T *__restrict__ u, const T *__restrict__ g, const T *__restrict__ tbrk,
    iw_t *iwhere, T *xcp, double tsum, double last_t, int64_t last_i, double *part) {
  double acc[1] = {0.0};
  const double poison = tsum * 0.0;  // 0, or NaN if tsum is not finite (see xcp_row in kernels_common.hpp)
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
      out[k] = xv[k] + poison;
      if (tb[k] >= 0.0) {
        const double d = -gv[k];
        if (done[k]) {
          if (d > 0.0) {
            out[k] = uv[k] + poison;
            iw[k] = 2;
          } else {
            out[k] = lv[k] + poison;
            iw[k] = 1;
          }
        } else if (tsum != 0.0) {
          out[k] = xv[k] + tsum * d;
        } else {
          out[k] = xv[k];
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
                       x, l, u, g, tbrk, iwhere, xcp, tsum, last_t, last_i, q.part());
    LB_LAUNCHED(q);
    launch_finalize(q, gr, 1, 0, 0);
  } else {
    hipLaunchKernelGGL((cauchy_finish_kernel<T, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n, row0,
                       x, l, u, g, tbrk, iwhere, xcp, tsum, last_t, last_i, q.part());
    LB_LAUNCHED(q);
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
  LB_LAUNCHED(q);
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_cauchy_scan<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, iw_t *, T *, WStore<T>, int, int); \
  template void launch_cauchy_window<T>(Queue &, int64_t, int64_t, const T *, double, int64_t, double, uint64_t *, uint32_t *, uint32_t, uint32_t *); \
  template void launch_cauchy_allkeys<T>(Queue &, int64_t, int64_t, const T *, double, int64_t, uint64_t *, uint32_t *); \
  template void launch_cauchy_gather<T>(Queue &, const uint32_t *, const uint64_t *, uint32_t, int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, const T *, const T *, Pend, double *); \
  template void launch_cauchy_gather_dyn<T>(Queue &, const uint32_t *, const uint64_t *, const uint32_t *, uint32_t, int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, const T *, const T *, Pend, double *); \
  template void launch_cauchy_window_fly<T>(Queue &, int64_t, int64_t, const T *, const T *, const T *, const nb_t *, const T *, const iw_t *, double, int64_t, double, uint64_t *, uint32_t *, uint32_t, uint32_t *, int); \
  template void launch_iwhere_update<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, iw_t *); \
  template void launch_tbrk_fill<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, const iw_t *, T *); \
  template void launch_cauchy_finish<T>(Queue &, int64_t, int64_t, const T *, const T *, const T *, const T *, const T *, iw_t *, T *, double, double, int64_t, int);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
