// k_pgcp.hip -- the opt-in parallel generalized-Cauchy-point search (LBFGSB_F_PARALLEL_GCP): the walk's
// state as prefix scans over the sorted breakpoints
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

#include <rocprim/rocprim.hpp>

namespace lbk {

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
    double *dd, double *a0, double *wb, double *uu, double *gi, int64_t row0,
    const uint64_t *__restrict__ lmask) {
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
      const int64_t off = (int64_t)((head - 1 + j) % m) * ldw + wrow(lmask, i);
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
// (C2 = compile-time bound of col2 = 2 col: the per-breakpoint vector w lives in registers -- with a
//  run-time bound it was a dynamically indexed local array, i.e. scratch memory)
template <int C2>
__global__ __launch_bounds__(BLOCK) void pgcp_terms_kernel(
    int64_t nb, int64_t nbp, int col2, double theta, const double *__restrict__ mm /* col2 x col2 */,
    const double *__restrict__ p0, const double *__restrict__ tt, const double *__restrict__ dd,
    const double *__restrict__ a0, const double *__restrict__ wb, const double *__restrict__ pp,
    const double *__restrict__ sq, double *df2, double *a1) {
  __shared__ double sm[C2 * C2];
  __shared__ double sp0[C2];
  for (int e = threadIdx.x; e < col2 * col2; e += blockDim.x) sm[e] = mm[e];
  for (int e = threadIdx.x; e < col2; e += blockDim.x) sp0[e] = p0[e];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += stride) {
    double w[C2];
#pragma unroll
    for (int c = 0; c < C2; ++c) w[c] = c < col2 ? wb[(int64_t)c * nbp + k] : 0.0;
    const double tk = tt[k];
    double wmc = 0.0, wmp = 0.0, wmw = 0.0;
    for (int a = 0; a < col2; ++a) {
      double y = 0.0;
#pragma unroll
      for (int b = 0; b < C2; ++b)
        if (b < col2) y += sm[a + b * col2] * w[b];
      const double pa = sp0[a] - pp[(int64_t)a * nbp + k];
      const double ca = tk * sp0[a] - sq[(int64_t)a * nbp + k];
      wmc += ca * y;
      wmp += pa * y;
      wmw += wb[(int64_t)a * nbp + k] * y;  // (= w[a], re-read: `a` is not a compile-time index)
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
                     q.part());
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
                     gi, row0, w.lmask);
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
#define LB_TERMS(C2V)                                                                                   \
  hipLaunchKernelGGL(pgcp_terms_kernel<C2V>, dim3(grid_for(nb, 1)), dim3(BLOCK), 0, q.stream, nb, nbp,   \
                     col2, theta, mm, p0, tt, dd, a0, wb, pp, sq, df2, a1)
  if (col2 <= 10)
    LB_TERMS(10);
  else if (col2 <= 20)
    LB_TERMS(20);
  else if (col2 <= 40)
    LB_TERMS(40);
  else
    LB_TERMS(2 * MAXM);
#undef LB_TERMS
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
                     q.part());
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

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_pgcp_gather<T>(Queue &, const uint32_t *, const uint64_t *, int64_t, int64_t, const T *, const T *, const T *, const T *, WStore<T>, int, int, double, const T *, const T *, Pend, double *, double *, double *, double *, double *, double *, int64_t); \
  template void launch_gcp_rest_mass<T>(Queue &, int64_t, const T *, const T *, double);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
