// k_cauchy.hip -- cauchy: n-loop, breakpoint selection/sort/gather, parallel GCP, finish; freev
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

#include <rocprim/rocprim.hpp>

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
                       nbd, g, iwhere, tbrk, w.ws, w.wy, w.zero, w.ld, w.m, head, col, q.d_part);
  } else {
    DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((cauchy_scan_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                          q.stream, n, x, l, u, nbd, g, iwhere, tbrk, w.ws, w.wy,
                                          w.zero, w.ld, w.m, head, col, q.d_part));
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

// =========================== parallel GCP search, col > 0 (opt-in) ============
// SURVEY.md 8f-2.  With the breakpoints sorted, the walk's state at breakpoint k is a prefix
// sum: p_k = p_0 - sum_{j<k} d_j wbp_j, c_k = t_k p_0 - sum_{j<=k} dt_j P_j, and the f1/f2
// recurrences (:1452-1481) become two more scans once the quadratic forms with M are known per
// breakpoint -- f2 with its clamp f2 = max(epsmch*f2_org, f2 + df2) (:1483) as a scan over the
// maps x -> max(B, x + A), which compose associatively.  Equal to the reference in exact
// arithmetic, not operation for operation: LBFGSB_F_PARALLEL_GCP only.  With several ranks each
// rank gathers the records of its own breakpoints (locally sorted), the records are all-gathered,
// merged by (t, global index) and every rank runs the same scans on all of them.
// Arrays are component-major: a[c * nbp + k], k = sorted position of the breakpoint.
template <typename T>
__global__ __launch_bounds__(BLOCK) void pgcp_gather_kernel(
    const uint32_t *__restrict__ idx, const uint64_t *__restrict__ keys, int64_t nb, int64_t nbp,
    const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const T *__restrict__ g, const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw,
    int m, int head, int col, double theta, const T *pr, const T *pd, Pend pe, double *tt,
    double *dd, double *a0, double *wb, double *uu, double *gi, int64_t row0) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    const int64_t i = idx[k];
    const double d = -(double)g[i];
    const double z = d > 0.0 ? (double)u[i] - (double)x[i] : (double)l[i] - (double)x[i];
    if (gi) gi[k] = (double)(row0 + i);
    tt[k] = __longlong_as_double((long long)keys[k]);
    dd[k] = d;
    a0[k] = d * d - theta * d * z;
    for (int j = 0; j < col; ++j) {
      const int64_t off = (int64_t)((head - 1 + j) % m) * ldw + i;
      const bool pj = pe.on && j == col - 1;
      const double yv = pj ? pend_y<T>((double)g[i], (double)pr[i]) : (double)wy[off];
      const double sv = theta * (pj ? pend_sx<T>((double)pd[i], (double)x[i], pe) : (double)ws[off]);
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
// f2 with its clamp (:1483) as an associative scan: crossing breakpoint k maps f2 to
// max(c, f2 + df2_k), c = epsmch*f2_org; maps x -> max(B, x + A) compose to
// (A1 + A2, max(B2, B1 + A2)).
struct F2Map {
  double a, b;
};
struct F2Compose {
  __host__ __device__ F2Map operator()(const F2Map &f, const F2Map &s) const {
    return F2Map{f.a + s.a, fmax(s.b, f.b + s.a)};
  }
};
__global__ __launch_bounds__(BLOCK) void pgcp_f2maps_kernel(int64_t nb, double cl,
                                                            const double *__restrict__ df2,
                                                            F2Map *maps) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride)
    maps[k] = F2Map{df2[k], cl};
}
// F2[k] = f2 after crossing breakpoint k
__global__ __launch_bounds__(BLOCK) void pgcp_f2apply_kernel(int64_t nb, double f2_0,
                                                             const F2Map *__restrict__ maps,
                                                             double *F2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride)
    F2[k] = fmax(maps[k].b, f2_0 + maps[k].a);
}
// df1_k = dt_k * f2_{k-1} + a1_k
__global__ __launch_bounds__(BLOCK) void pgcp_f1_kernel(int64_t nb, double f2_0,
                                                        const double *__restrict__ tt,
                                                        const double *__restrict__ F2,
                                                        const double *__restrict__ a1, double *df1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    const double dt = tt[k] - (k > 0 ? tt[k - 1] : 0.0);
    const double f2p = k > 0 ? F2[k - 1] : f2_0;
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
    const double f2p = k > 0 ? sf2[k - 1] : f2_0;  // (F2: the clamped f2 itself)
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
                                 const double *__restrict__ gi, double *out) {
  const int c = threadIdx.x;
  if (c == 0) {
    out[0] = ks > 0 ? tt[ks - 1] : 0.0;
    out[1] = f1_0 + (ks > 0 ? sf1[ks - 1] : 0.0);
    out[2] = ks > 0 ? sf2[ks - 1] : f2_0;
    // row of the last crossed breakpoint: global (gi, merged multi-rank order) or local (idx)
    out[3] = ks > 0 ? (gi ? gi[ks - 1] : (double)idx[ks - 1]) : -1.0;
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

// ---- several ranks: merge of the all-gathered, per-rank sorted records ----
// G holds, per rank, `narr` arrays of nbp doubles (array 0 = tt).  keys/vals: one slot per
// (rank, k); slots beyond a rank's count sort to the end.
__global__ __launch_bounds__(BLOCK) void pgcp_mergekeys_kernel(int nranks, int64_t nbp, int narr,
                                                               const double *__restrict__ counts,
                                                               const double *__restrict__ G,
                                                               uint64_t *keys, uint32_t *vals) {
  const int64_t total = (int64_t)nranks * nbp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
    const int rk = (int)(s / nbp);
    const int64_t k = s - (int64_t)rk * nbp;
    const bool live = (double)k < counts[rk];
    keys[s] = live ? (uint64_t)__double_as_longlong(G[(int64_t)rk * narr * nbp + k]) : ~0ull;
    vals[s] = (uint32_t)s;
  }
}
// out arrays (stride NBp) <- gathered arrays in merged order.  Array a of G goes to out + map[a]*NBp
__global__ __launch_bounds__(BLOCK) void pgcp_permute_kernel(int64_t NB, int64_t NBp, int64_t nbp,
                                                             int narr, const uint32_t *__restrict__ vals,
                                                             const double *__restrict__ G, double *out,
                                                             const int *__restrict__ map) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < NB; k += stride) {
    const int64_t s = vals[k];
    const int64_t rk = s / nbp, kk = s - rk * nbp;
    const double *src = G + rk * narr * nbp + kk;
    for (int a = 0; a < narr; ++a) out[(int64_t)map[a] * NBp + k] = src[(int64_t)a * nbp];
  }
}
void launch_pgcp_mergekeys(Queue &q, int nranks, int64_t nbp, int narr, const double *counts,
                           const double *G, uint64_t *keys, uint32_t *vals) {
  hipLaunchKernelGGL(pgcp_mergekeys_kernel, dim3(grid_for((int64_t)nranks * nbp, 1)), dim3(BLOCK), 0,
                     q.stream, nranks, nbp, narr, counts, G, keys, vals);
  LB_LAUNCHED(q);
}
void launch_pgcp_permute(Queue &q, int64_t NB, int64_t NBp, int64_t nbp, int narr, const uint32_t *vals,
                         const double *G, double *out, const int *map) {
  hipLaunchKernelGGL(pgcp_permute_kernel, dim3(grid_for(NB, 1)), dim3(BLOCK), 0, q.stream, NB, NBp, nbp,
                     narr, vals, G, out, map);
  LB_LAUNCHED(q);
}
// f2 through all breakpoints with the clamp: df2 (in) -> F2 (out, may alias df2); maps = 2 nb doubles
size_t f2scan_temp_bytes(size_t count) {
  size_t b = 0;
  (void)rocprim::deterministic_inclusive_scan(nullptr, b, (const F2Map *)nullptr, (F2Map *)nullptr, count,
                                              F2Compose(), (hipStream_t)0);
  return b;
}
void launch_pgcp_f2(Queue &q, void *d_temp, size_t temp_bytes, int64_t nb, double f2_0, double cl,
                    const double *df2, double *maps, double *F2) {
  F2Map *mp = reinterpret_cast<F2Map *>(maps);
  hipLaunchKernelGGL(pgcp_f2maps_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, cl, df2, mp);
  (void)rocprim::deterministic_inclusive_scan(d_temp, temp_bytes, mp, mp, (size_t)nb, F2Compose(), q.stream);
  hipLaunchKernelGGL(pgcp_f2apply_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, f2_0, mp, F2);
  q.launches += 3;
}
// the part of d'd that is still moving beyond t*: rows whose breakpoint lies after it (or that
// never reach a bound).  The closed-form GCP (col = 0) is valid only while this stays above
// epsmch * d'd -- below it the reference's clamp f2 = max(epsmch*f2_org, f2) (:1483) takes over.
template <typename T>
__global__ __launch_bounds__(BLOCK) void gcp_rest_mass_kernel(int64_t n, const T *__restrict__ g,
                                                              const T *__restrict__ tbrk, double tstar,
                                                              double *part) {
  double acc[1] = {0.0};
  for_rows<T>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double gv[W], tb[W];
    ld<W>(g + i, gv);
    ld<W>(tbrk + i, tb);
#pragma unroll
    for (int k = 0; k < W; ++k)
      if (tb[k] > tstar) acc[0] = acc[0] + gv[k] * gv[k];
  });
  block_reduce_store<1>(acc, 1, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_gcp_rest_mass(Queue &q, int64_t n, const T *g, const T *tbrk, double tstar) {
  const int gr = grid_for(n, VecOf<T>::V);
  hipLaunchKernelGGL(gcp_rest_mass_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, g, tbrk, tstar,
                     q.d_part);
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 1, 0, 0);
}

// (the bitwise-reproducible variants: with several ranks every rank runs these scans on the same
//  data and must arrive at the same bits; the default look-back scan groups its partial sums
//  by timing)
size_t scan_temp_bytes(size_t count) {
  size_t b1 = 0, b2 = 0;
  (void)rocprim::deterministic_inclusive_scan(nullptr, b1, (const double *)nullptr, (double *)nullptr, count,
                                              rocprim::plus<double>(), (hipStream_t)0);
  (void)rocprim::deterministic_exclusive_scan(nullptr, b2, (const double *)nullptr, (double *)nullptr, 0.0,
                                              count, rocprim::plus<double>(), (hipStream_t)0);
  return b1 > b2 ? b1 : b2;
}
void launch_scan(Queue &q, void *d_temp, size_t temp_bytes, const double *in, double *out,
                 size_t count, int exclusive) {
  if (exclusive)
    (void)rocprim::deterministic_exclusive_scan(d_temp, temp_bytes, in, out, 0.0, count,
                                                rocprim::plus<double>(), q.stream);
  else
    (void)rocprim::deterministic_inclusive_scan(d_temp, temp_bytes, in, out, count, rocprim::plus<double>(),
                                                q.stream);
  LB_LAUNCHED(q);
}
template <typename T>
void launch_pgcp_gather(Queue &q, const uint32_t *idx, const uint64_t *keys, int64_t nb, int64_t nbp,
                        const T *x, const T *l, const T *u, const T *g, WStore<T> w, int head, int col,
                        double theta, const T *pr, const T *pd, Pend pe, double *tt, double *dd,
                        double *a0, double *wb, double *uu, double *gi, int64_t row0) {
  const int gr = grid_for(nb, 1);
  hipLaunchKernelGGL(pgcp_gather_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, idx, keys, nb, nbp, x,
                     l, u, g, w.ws, w.wy, w.ld, w.m, head, col, theta, pr, pd, pe, tt, dd, a0, wb, uu,
                     gi, row0);
  LB_LAUNCHED(q);
}
void launch_pgcp_last(Queue &q, int64_t nb, int64_t nbp, int col2, const double *uu, double *uu_last) {
  hipLaunchKernelGGL(pgcp_last_kernel, dim3(1), dim3(64), 0, q.stream, nb, nbp, col2, uu, uu_last);
  LB_LAUNCHED(q);
}
void launch_pgcp_dtp(Queue &q, int64_t nb, int64_t nbp, int col2, const double *tt, const double *pp,
                     double *qq) {
  hipLaunchKernelGGL(pgcp_dtp_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, nbp, col2,
                     tt, pp, qq);
  LB_LAUNCHED(q);
}
void launch_pgcp_terms(Queue &q, int64_t nb, int64_t nbp, int col2, double theta, const double *mm,
                       const double *p0, const double *tt, const double *dd, const double *a0,
                       const double *wb, const double *pp, const double *sq, double *df2, double *a1) {
  hipLaunchKernelGGL(pgcp_terms_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, nbp, col2,
                     theta, mm, p0, tt, dd, a0, wb, pp, sq, df2, a1);
  LB_LAUNCHED(q);
}
void launch_pgcp_f1(Queue &q, int64_t nb, double f2_0, const double *tt, const double *sf2,
                    const double *a1, double *df1) {
  hipLaunchKernelGGL(pgcp_f1_kernel, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, f2_0, tt, sf2,
                     a1, df1);
  LB_LAUNCHED(q);
}
void launch_pgcp_find(Queue &q, int64_t nb, double f1_0, double f2_0, const double *tt,
                      const double *sf1, const double *sf2) {
  const int gr = grid_for(nb, 1);
  hipLaunchKernelGGL(pgcp_find_kernel, dim3(gr), dim3(BLOCK), 0, q.stream, nb, f1_0, f2_0, tt, sf1, sf2,
                     q.d_part);
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 0, 1, 0);
}
void launch_pgcp_pick(Queue &q, int64_t ks, int64_t nb, int64_t nbp, int col2, double f1_0, double f2_0,
                      const double *tt, const double *sf1, const double *sf2, const double *pp,
                      const double *uu_last, const double *sq, const uint32_t *idx, const double *gi,
                      double *out) {
  hipLaunchKernelGGL(pgcp_pick_kernel, dim3(1), dim3(64), 0, q.stream, ks, nb, nbp, col2, f1_0, f2_0, tt,
                     sf1, sf2, pp, uu_last, sq, idx, gi, out);
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

size_t sort_pairs_temp_bytes(size_t count) {
  size_t b1 = 0, b2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b1, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, count, 0, 64,
                                  (hipStream_t)0);
  (void)rocprim::radix_sort_pairs(nullptr, b2, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (const uint64_t *)nullptr, (uint64_t *)nullptr, count, 0, 32,
                                  (hipStream_t)0);
  size_t b3 = 0;
  (void)rocprim::radix_sort_keys(nullptr, b3, (const uint32_t *)nullptr, (uint32_t *)nullptr, count, 0, 32,
                                 (hipStream_t)0);
  b1 = b1 > b2 ? b1 : b2;
  return b1 > b3 ? b1 : b3;
}
// ---- ascending order for a list of <= 2^18 row numbers (freev's changed rows): the list is
//      appended with an atomic counter, i.e. in an order that may change from run to run, and
//      formk's patch sums run over it -- sorted, the sums are reproducible bit for bit ----
constexpr int SMALL_SORT = 2048;
__global__ __launch_bounds__(BLOCK) void sort_u32_small_kernel(uint32_t *keys, uint32_t cnt, int npow2) {
  __shared__ uint32_t sm[SMALL_SORT];
  // (the network is only as large as the list: npow2 = the power of two >= cnt, <= SMALL_SORT)
  for (int k = threadIdx.x; k < npow2; k += BLOCK) sm[k] = (uint32_t)k < cnt ? keys[k] : 0xFFFFFFFFu;
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1)      // bitonic network, one workgroup
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int k = threadIdx.x; k < npow2 / 2; k += BLOCK) {
        const int lo = 2 * k - (k & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint32_t a = sm[lo], b = sm[hi];
        if ((a > b) == up) sm[lo] = b, sm[hi] = a;
      }
      __syncthreads();
    }
  for (int k = threadIdx.x; k < npow2; k += BLOCK)
    if ((uint32_t)k < cnt) keys[k] = sm[k];
}
uint32_t *launch_sort_u32(Queue &q, void *d_temp, size_t temp_bytes, uint32_t *keys, uint32_t *scratch,
                          uint32_t count) {
  if (count <= 1) return keys;
  if (count <= (uint32_t)SMALL_SORT) {
    int npow2 = 2;
    while ((uint32_t)npow2 < count) npow2 <<= 1;
    hipLaunchKernelGGL(sort_u32_small_kernel, dim3(1), dim3(BLOCK), 0, q.stream, keys, count, npow2);
    LB_LAUNCHED(q);
    return keys;
  }
  (void)rocprim::radix_sort_keys(d_temp, temp_bytes, keys, scratch, (size_t)count, 0, 32, q.stream);
  LB_LAUNCHED(q);
  return scratch;
}
// ---- several ranks: the all-gathered record chunks (one sorted run per rank) merged ON THE DEVICE ----
// all = nranks blocks of `stride` doubles: { count, more, records[chunk][recl] }, every block in
// (t, global index) order.  A stable sort on t of the concatenation in rank order IS the (t, global
// index) order of the union (ranks own ascending row blocks).  out = { header[4 nranks] = count, more,
// t and index of the last record of every rank | merged records | one byte per merged record: its rank }.
__global__ __launch_bounds__(BLOCK) void merge_keys_kernel(const double *__restrict__ all, int nranks,
                                                           uint32_t chunk, int recl, size_t stride,
                                                           uint64_t *keys, uint32_t *vals, double *out) {
  const size_t S = (size_t)nranks * chunk;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (size_t)gridDim.x * blockDim.x) {
    const int rk = (int)(s / chunk);
    const uint32_t k = (uint32_t)(s % chunk);
    const double *base = all + (size_t)rk * stride;
    const uint32_t lr = (uint32_t)base[0];
    keys[s] = k < lr ? (uint64_t)__double_as_longlong(base[2 + (size_t)k * recl]) : ~0ull;
    vals[s] = (uint32_t)s;
    if (k == 0) {
      out[4 * rk + 0] = base[0];
      out[4 * rk + 1] = base[1];
      out[4 * rk + 2] = lr ? base[2 + (size_t)(lr - 1) * recl] : 0.0;
      out[4 * rk + 3] = lr ? base[2 + (size_t)(lr - 1) * recl + 1] : 0.0;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void merge_permute_kernel(const double *__restrict__ all, int nranks,
                                                              uint32_t chunk, int recl, size_t stride,
                                                              const uint64_t *__restrict__ keys,
                                                              const uint32_t *__restrict__ vals, double *out) {
  const size_t S = (size_t)nranks * chunk;
  double *recs = out + 4 * (size_t)nranks;
  unsigned char *rb = reinterpret_cast<unsigned char *>(recs + S * (size_t)recl);
  const size_t total = S * (size_t)recl;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t p = e / recl;
    const int f = (int)(e % recl);
    if (keys[p] == ~0ull) continue;  // (slots beyond the records the ranks sent)
    const uint32_t s = vals[p];
    const int rk = (int)(s / chunk);
    const uint32_t k = s % chunk;
    recs[e] = all[(size_t)rk * stride + 2 + (size_t)k * recl + f];
    if (f == 0) rb[p] = (unsigned char)rk;
  }
}
void launch_merge_chunks(Queue &q, int nranks, uint32_t chunk, int recl, size_t stride, const double *all,
                         uint64_t *keys0, uint64_t *keys1, uint32_t *vals0, uint32_t *vals1, void *d_temp,
                         size_t temp_bytes, double *out) {
  const size_t S = (size_t)nranks * chunk;
  hipLaunchKernelGGL(merge_keys_kernel, dim3(grid_for((int64_t)S, 1)), dim3(BLOCK), 0, q.stream, all, nranks,
                     chunk, recl, stride, keys0, vals0, out);
  LB_LAUNCHED(q);
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, keys0, keys1, vals0, vals1, S, 0, 64, q.stream);
  LB_LAUNCHED(q);
  hipLaunchKernelGGL(merge_permute_kernel, dim3(grid_for((int64_t)(S * recl), 4)), dim3(BLOCK), 0, q.stream, all,
                     nranks, chunk, recl, stride, keys1, vals1, out);
  LB_LAUNCHED(q);
}
void launch_sort_by_idx(Queue &q, void *d_temp, size_t temp_bytes, const uint32_t *idx_in,
                        uint32_t *idx_out, const uint64_t *keys_in, uint64_t *keys_out,
                        size_t count) {
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, idx_in, idx_out, keys_in, keys_out, count, 0,
                                  32, q.stream);
  LB_LAUNCHED(q);
}
void launch_sort_pairs(Queue &q, void *d_temp, size_t temp_bytes, const uint64_t *keys_in,
                       uint64_t *keys_out, const uint32_t *idx_in, uint32_t *idx_out,
                       size_t count) {
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, count, 0,
                                  64, q.stream);
  LB_LAUNCHED(q);
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
              ? pend_sx<T>((double)pd[i], (double)x[i], pe)
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
              ? pend_sx<T>((double)pd[i], (double)x[i], pe)
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
                     row0, x, l, u, g, w.ws, w.wy, w.ld, w.m, head, col, pr, pd, pe, rec);
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
                       x, l, u, g, tbrk, iwhere, xcp, tsum, last_t, last_i, q.d_part);
    LB_LAUNCHED(q);
    launch_finalize(q, gr, 1, 0, 0);
  } else {
    hipLaunchKernelGGL((cauchy_finish_kernel<T, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n, row0,
                       x, l, u, g, tbrk, iwhere, xcp, tsum, last_t, last_i, q.d_part);
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

// =========================== freev (:1980-2059) ==============================
__global__ __launch_bounds__(BLOCK) void freev_count_kernel(int64_t n,
                                                            const iw_t *__restrict__ iwhere,
                                                            int8_t *wasfree, double *part,
                                                            uint32_t *chg, uint32_t chg_cap,
                                                            uint32_t *chg_count) {
  // 16 rows per lane and trip (one 16-byte load of each byte array).  Rows whose status changed
  // are collected per workgroup in LDS and appended to the global list with ONE global atomic
  // per flush (a same-address atomic per row would serialise: 1e5 changes x ~12 ns)
  constexpr int R = 16, LCAP = 8192;
  __shared__ uint32_t lbuf[LCAP];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  double acc[3] = {0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * R;
  const int64_t ntrip = (n + stride - 1) / stride;  // uniform trip count (barriers inside)
  for (int64_t trip = 0; trip < ntrip; ++trip) {
    const int64_t i = trip * stride + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * R;
    if (i < n) {
      typedef int v4i __attribute__((ext_vector_type(4)));
      union {
        v4i v;
        int8_t b[R];
      } iw, wf;
      const bool full = i + R <= n;  // (both arrays are allocated with 32 spare elements, but
                                     //  rows beyond n must neither be counted nor written)
      if (full) {
        iw.v = *reinterpret_cast<const v4i *>(iwhere + i);
        wf.v = *reinterpret_cast<const v4i *>(wasfree + i);
      } else {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          iw.b[k] = i + k < n ? iwhere[i + k] : (iw_t)1;
          wf.b[k] = i + k < n ? wasfree[i + k] : (int8_t)0;
        }
      }
      int nfr = 0, nen = 0, nlv = 0;
      unsigned changed = 0;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const bool fr = iw.b[k] <= 0, was = wf.b[k] != 0;
        nfr += fr, nen += fr && !was, nlv += !fr && was;
        changed |= (fr != was) ? (1u << k) : 0u;
        wf.b[k] = fr ? 1 : 0;
      }
      acc[0] += nfr, acc[1] += nen, acc[2] += nlv;
      if (changed) {  // (few rows: keeps the pass that follows free of drained store traffic)
        if (chg) {
          const uint32_t pos = atomicAdd(&lcount, (uint32_t)__builtin_popcount(changed));  // LDS atomic
          uint32_t w = pos;
#pragma unroll
          for (int k = 0; k < R; ++k)
            if ((changed >> k) & 1u) lbuf[w++] = (uint32_t)(i + k) | (wf.b[k] ? 0u : 0x80000000u);
        }
        if (full) {
          *reinterpret_cast<v4i *>(wasfree + i) = wf.v;
        } else {
#pragma unroll
          for (int k = 0; k < R; ++k)
            if (i + k < n) wasfree[i + k] = wf.b[k];
        }
      }
    }
    if (chg) {
      __syncthreads();
      const uint32_t cnt = lcount;
      if (cnt > LCAP - BLOCK * R || trip == ntrip - 1) {  // uniform: flush
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
  const int gr = grid_for(n, 16);
  if (chg) (void)hipMemsetAsync(chg_count, 0, sizeof(uint32_t), q.stream);
  hipLaunchKernelGGL(freev_count_kernel, dim3(gr), dim3(BLOCK), 0, q.stream, n, iwhere, wasfree,
                     q.d_part, chg, chg_cap, chg_count);
  LB_LAUNCHED(q);
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


// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_cauchy_scan<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, iw_t *, T *, WStore<T>, int, int); \
  template void launch_cauchy_window<T>(Queue &, int64_t, int64_t, const T *, double, int64_t, double, uint64_t *, uint32_t *, uint32_t, uint32_t *); \
  template void launch_cauchy_allkeys<T>(Queue &, int64_t, int64_t, const T *, double, int64_t, uint64_t *, uint32_t *); \
  template void launch_cauchy_gather<T>(Queue &, const uint32_t *, const uint64_t *, uint32_t, int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, const T *, const T *, Pend, double *); \
  template void launch_cauchy_gather_dyn<T>(Queue &, const uint32_t *, const uint64_t *, const uint32_t *, uint32_t, int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, const T *, const T *, Pend, double *); \
  template void launch_cauchy_window_fly<T>(Queue &, int64_t, int64_t, const T *, const T *, const T *, const nb_t *, const T *, const iw_t *, double, int64_t, double, uint64_t *, uint32_t *, uint32_t, uint32_t *, int); \
  template void launch_iwhere_update<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, iw_t *); \
  template void launch_pgcp_gather<T>(Queue &, const uint32_t *, const uint64_t *, int64_t, int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, double, const T *, const T *, Pend, double *, double *, double *, double *, double *, double *, int64_t); \
  template void launch_gcp_rest_mass<T>(Queue &, int64_t, const T *, const T *, double); \
  template void launch_tbrk_fill<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, const iw_t *, T *); \
  template void launch_cauchy_finish<T>(Queue &, int64_t, int64_t, const T *, const T *, const T *, const T *, const T *, iw_t *, T *, double, double, int64_t, int);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
