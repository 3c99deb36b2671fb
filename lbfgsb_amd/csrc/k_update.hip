// k_update.hip -- matupd and its fusion with cauchy's n-loop
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

namespace lbk {

// =========================== mainlb :812-824 + matupd (:2291-2346) ===========
// ncol_old = col - 1 older pairs (logical order from head); new pair goes to
// physical column itail (1-based).
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void update_pairs_kernel(
    int64_t n, const T *__restrict__ g, const T *__restrict__ r, const T *__restrict__ d,
    double stp, T *ws, T *wy, const T *__restrict__ zero, int64_t ldw, int m, int head, int nold,
    int itail, double *part) {
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
      const int64_t off = col_off(j, nold, head, m, ldw) + i;
      ld_col<T, W, NT>(j < nold, wy + off, zero, a[j]);
      ld_col<T, W, NT>(j < nold, ws + off, zero, b[j]);
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
                                         q.stream, n, g, r, d, stp, w.ws, w.wy, w.zero, w.ld, w.m,
                                         head, nold, itail, q.d_part));
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
template <typename T>
struct UpdScanCtx {
  const T *x, *l, *u, *g, *r, *d, *ws, *wy, *zero;
  const int32_t *nbd;
  const iw_t *iwhere;
  int64_t ldw;
  int m, head, nold;
};
template <typename T, int MC, int W, bool NT>
struct UpdScanTrip {
  static constexpr int NL = 8 + 2 * MC;
  RawOf<T, W> rx, rl, ru, rg, rr, rd, ra[MC], rb[MC];
  RawOf<int32_t, W> rnb;
  RawOf<iw_t, W> riw;
  __device__ __forceinline__ void issue(const UpdScanCtx<T> &c, int64_t i) {
    constexpr int B = (int)sizeof(T) * W;
    raw_issue<B, NT>(rx, c.x + i);
    raw_issue<B, NT>(rl, c.l + i);
    raw_issue<B, NT>(ru, c.u + i);
    raw_issue<B, NT>(rg, c.g + i);
    raw_issue<B, NT>(rr, c.r + i);
    raw_issue<B, NT>(rd, c.d + i);
    raw_issue<4 * W, false>(rnb, c.nbd + i);
    raw_issue<W, false>(riw, c.iwhere + i);
    issue_cols<T, MC, W, NT>(c.wy, c.ws, (const T *)nullptr, (const T *)nullptr, c.zero, i, c.nold, c.head,
                             c.m, c.ldw, Pend{0, 1.0}, ra, rb);
  }
  __device__ __forceinline__ void land() {
    raw_land(rx);
    raw_land(rl);
    raw_land(ru);
    raw_land(rg);
    raw_land(rr);
    raw_land(rd);
    raw_land(rnb);
    raw_land(riw);
    land_cols<T, MC, W>(ra, rb);
  }
};
template <typename T, int MC, bool NT, bool PIPE>
__global__ __launch_bounds__(BLOCK) void update_scan_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const int32_t *__restrict__ nbd, const T *__restrict__ g, const T *__restrict__ r,
    const T *__restrict__ d, double stp, iw_t *iwhere, T *tbrk, T *ws, T *wy,
    const T *__restrict__ zero, int64_t ldw, int m, int head, int nold, int itail, int store_pair,
    int store_iw, double *part) {
  constexpr int NA = 4 * MC + 11;
  constexpr int V = RowsPerAcc<T, MC, NA>::V;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  acc[4 * MC + 9] = LB_INF;
  const int64_t offn = (int64_t)(itail - 1) * ldw;
  const UpdScanCtx<T> ctx{x, l, u, g, r, d, ws, wy, zero, nbd, iwhere, ldw, m, head, nold};
  for_rows_raw<UpdScanTrip<T, MC, V, NT>, UpdScanTrip<T, MC, 1, NT>, V, PIPE, 0>(
      n, ctx, [&](auto &tr, int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], rv[W], dv[W], tb[W], ng[W], a[MC][W], b[MC][W];
    int nb[W], iw[W];
    raw_get<W>(tr.rx, (const T *)nullptr, xv);
    raw_get<W>(tr.rl, (const T *)nullptr, lv);
    raw_get<W>(tr.ru, (const T *)nullptr, uv);
    raw_get<W>(tr.rg, (const T *)nullptr, gv);
    raw_get<W>(tr.rr, (const T *)nullptr, rv);
    raw_get<W>(tr.rd, (const T *)nullptr, dv);
    raw_geti<W>(tr.rnb, (const int32_t *)nullptr, nb);
    raw_geti<W>(tr.riw, (const iw_t *)nullptr, iw);
    get_cols<T, MC, W>(tr.ra, tr.rb, a, b);
    bool iw_changed = false;
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
  DISPATCH_MAXC_NT(nold, q.nt,
                   DISPATCH_PIPE(MC, hipLaunchKernelGGL((update_scan_kernel<T, MC, NTV, PIPEV>), dim3(gr),
                                                        dim3(BLOCK), 0, q.stream, n, x, l, u, nbd, g, r, d,
                                                        stp, iwhere, tbrk, w.ws, w.wy, w.zero, w.ld, w.m,
                                                        head, nold, itail, store_pair, store_iw, q.d_part)));
  q.launches++;
  launch_finalize(q, gr, 4 * maxc_for(nold) + 9, 1, 1);
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_update_pairs<T>(Queue &, int64_t, const T *, const T *, const T *, double, WStore<T>, int, int, int); \
  template void launch_update_scan<T>(Queue &, int64_t, const T *, const T *, const T *, const int32_t *, const T *, const T *, const T *, double, iw_t *, T *, WStore<T>, int, int, int, int, int);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
