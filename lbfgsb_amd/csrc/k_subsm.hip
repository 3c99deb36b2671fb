// k_subsm.hip -- subsm: Newton direction, projected step, line-search set-up, pair commit
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

namespace lbk {

// =========================== subsm (:2676-2885) ==============================
// Newton direction of one free row (cmprlb :1560-1583 then subsm :2770-2780): the reduced
// gradient r is recomputed here exactly as cmprlb_wtv_kernel computed it for W'r.
template <int MC, bool FULL = false>
__device__ __forceinline__ double subsm_dir(double xk, double zk, double gk, const double (&a)[MC],
                                            const double (&b)[MC], int col, double theta,
                                            double rtheta, const Coef &cf, const Coef &wv) {
  // (unconstrained shortcut of cmprlb, r = -g :1560-1563: zk == xk and cf == 0 give exactly that)
  double dk = -theta * (zk - xk) - gk;
#pragma unroll
  for (int j = 0; j < MC; ++j)
    if (FULL || j < col) dk = dk + a[j] * cf.a[j] + b[j] * cf.a[MAXM + j];
#pragma unroll
  for (int j = 0; j < MC; ++j)
    if (FULL || j < col) dk = dk + a[j] * wv.a[j] / theta + b[j] * wv.a[MAXM + j];
  return rtheta * dk;  // dscal (:2780)
}

// One pass: Newton direction, projected step (:2789-2816), dd_p (:2824-2827) and -- because the
// projected point is final unless the rare backtracking branch (:2830-2879) is taken -- what
// mainlb :720-722 and the first call of lnsrlb (:2196-2236) do next: d = z - x, t = x, r = g,
// dtd = d'd, the stpmx ratios; g'd is dd_p itself.  The Cauchy point is evaluated per row
// (xcp_row), the subspace minimiser written to `zout`; neither xp (:2787) nor the direction is
// stored (the backtracking branch regenerates both: cauchy_finish_kernel, subsm_dir_kernel).
// pr / pd: where a pending pair's y and s are read from (r, and d or t); rout / tvec: where r = g
// and t = x are stored (nullptr: not stored, see below).
// res: sum [0] = #bound hits (iword), [1] = dd_p (= g'd), [2] = dtd ; min [3] = stpmx
template <typename T>
struct SubsmCtx {
  const T *l, *u, *xx, *gg, *ws, *wy, *zero, *r, *pd;
  const nb_t *nbd;
  const iw_t *iwhere;
  int64_t ldw;
  int m, head, col;
  Pend pe;
  int ub;  // uniform bounds: bit 0 l, bit 1 u, bit 2 nbd are 64-byte constant buffers (see UpdScanCtx)
};
template <typename T, int MC, int W, bool NT, bool PSPEC>
struct SubsmTrip {
  static constexpr int NL = 6 + 2 * MC;
  static constexpr bool CW = false;
  __device__ __forceinline__ int64_t wrow(int64_t i) const { return i; }  // where row i sits in a W column
  RawOf<T, W> rl, ru, rx, rg, ra[MC], rb[MC];
  RawOf<nb_t, W> rnb;
  RawOf<iw_t, W> riw;
  __device__ __forceinline__ void issue(const SubsmCtx<T> &c, int64_t i) {
    constexpr int B = (int)sizeof(T) * W;
    raw_issue<B, NT>(rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(rx, c.xx + i);
    raw_issue<B, NT>(rg, c.gg + i);
    raw_issue<W, false>(rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<W, false>(riw, c.iwhere + i);
    // a pending pair is read from (r, d) -- or (r, t) when d is implicit -- which this pass
    // overwrites further down
    issue_cols<T, MC, W, NT, PSPEC>(c.wy, c.ws, c.r, c.pd, c.zero, i, c.col, c.head, c.m, c.ldw, c.pe,
                                    ra, rb);
  }
  __device__ __forceinline__ void land() {
    raw_land(rl);
    raw_land(ru);
    raw_land(rx);
    raw_land(rg);
    raw_land(rnb);
    raw_land(riw);
    land_cols<T, MC, W>(ra, rb);
  }
};
// One row under the tile-local free-row layout of W (for_tiles_cw): x, g, the bounds, iwhere and a pending
// pair's vectors from row i, the stored columns from the row's slot if its layout bit is set -- only free rows
// (iwhere <= 0) use them (cmprlb :1565-1583 and subsm :2770-2778 run over Index(1:nfree)); a free row whose bit is
// clear fetches them in the kernel body (reload_cols).
template <typename T, int MC, bool NT, bool PSPEC>
struct SubsmTripCW : SubsmTrip<T, MC, 1, NT, PSPEC> {
  static constexpr bool CW = true;
  int64_t ws_;
  bool lf_;
  __device__ __forceinline__ int64_t wrow(int64_t) const { return ws_; }
  template <bool NTL>
  __device__ __forceinline__ void cols(const SubsmCtx<T> &c, int64_t i, bool lf) {
    constexpr int B = (int)sizeof(T);
    if constexpr (PSPEC) {
#pragma unroll
      for (int j = 0; j < MC - 1; ++j) {
        const int64_t off = (int64_t)((c.head - 1 + j) % c.m) * c.ldw + ws_;
        raw_issue<B, NTL>(this->ra[j], lf ? c.wy + off : c.zero);
        raw_issue<B, NTL>(this->rb[j], lf ? c.ws + off : c.zero);
      }
      raw_issue<B, NTL>(this->ra[MC - 1], c.r + i);
      raw_issue<B, NTL>(this->rb[MC - 1], c.pd + i);
    } else {
#pragma unroll
      for (int j = 0; j < MC; ++j) {
        const int64_t off = col_off(j, c.col, c.head, c.m, c.ldw) + ws_;
        const bool live = j < c.col, pj = c.pe.on && j == c.col - 1;
        const T *py = pj ? c.r + i : c.wy + off, *ps = pj ? c.pd + i : c.ws + off;
        raw_issue<B, NTL>(this->ra[j], live && (lf || pj) ? py : c.zero);
        raw_issue<B, NTL>(this->rb[j], live && (lf || pj) ? ps : c.zero);
      }
    }
  }
  __device__ __forceinline__ void issue_cw(const SubsmCtx<T> &c, int64_t i, int64_t slot, bool lf) {
    constexpr int B = (int)sizeof(T);
    ws_ = slot, lf_ = lf;
    ri_ = i;
    raw_issue<B, NT>(this->rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(this->ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(this->rx, c.xx + i);
    raw_issue<B, NT>(this->rg, c.gg + i);
    raw_issue<1, false>(this->rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<1, false>(this->riw, c.iwhere + i);
    cols<NT>(c, i, lf);
  }
  int64_t ri_;
  __device__ __forceinline__ void reload_cols(const SubsmCtx<T> &c) { cols<false>(c, ri_, true); }
};
// CW: W in the tile-local free-row layout `lmask` (fp64, MC <= 10; for_tiles_cw)
template <typename T, int MC, bool NT, bool PSPEC, bool PIPE, bool CW = false>
__global__ __launch_bounds__(BLOCK) void subsm_update_kernel(
    int64_t n, double tsum, T *__restrict__ zout, const T *pr, T *rout,
    const T *__restrict__ l, const T *__restrict__ u, const nb_t *__restrict__ nbd,
    const iw_t *__restrict__ iwhere, const T *xx, const T *__restrict__ gg,
    const T *__restrict__ ws, const T *__restrict__ wy, const T *__restrict__ zero, int64_t ldw,
    int m, int head, int col, double theta, Coef cf, Coef wv, T *dvec, T *tvec,
    T *xout, int do_stpmx, Pend pe, const T *pd, T *cwy, T *cws, int ub, double *part, int pstride,
    const uint64_t *__restrict__ lmask = nullptr) {
  static_assert(!CW || (!PIPE && sizeof(T) == 8 && MC <= 10), "compact W: fp64, MC <= 10, one trip in flight");
  double acc[4] = {0.0, 0.0, 0.0, 1.0e10};
  const double rtheta = 1.0 / theta;
  constexpr int V = RowsPer<T, MC>::V;
  // stores every trip issues (a lower bound: the counted wait of the pipelined loop may then wait
  // for a few stores too): the trial point / z (+ the committed pair in the steady-state shape)
  constexpr int NS = PSPEC ? 3 : 1;
  const SubsmCtx<T> ctx{l, u, xx, gg, ws, wy, zero, pr, pd, nbd, iwhere, ldw, m, head, col, pe, ub};
  __shared__ T dict[16];
  dict_fill<T>(dict, l, u, ub);
  auto body = [&](auto &tr, int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double zv[W], lv[W], uv[W], xv[W], gv[W], a[MC][W], b[MC][W];
    int nb[W], iw[W];
    raw_get<W>(tr.rl, (const T *)nullptr, lv);
    raw_get<W>(tr.ru, (const T *)nullptr, uv);
    raw_get<W>(tr.rx, (const T *)nullptr, xv);
    raw_get<W>(tr.rg, (const T *)nullptr, gv);
    raw_geti<W>(tr.rnb, (const nb_t *)nullptr, nb);
    raw_geti<W>(tr.riw, (const iw_t *)nullptr, iw);
    dict_apply<T, W>(dict, ub, nb, lv, uv);
    if constexpr (std::remove_reference_t<decltype(tr)>::CW) {
      // a free row whose layout bit is clear (it became free after the layout was made)
      const bool miss = !tr.lf_ && iw[0] <= 0;
      if (__ballot(miss) != 0ull) {
        if (miss) tr.reload_cols(ctx);
      }
    }
    get_cols<T, MC, W>(tr.ra, tr.rb, a, b);
    fix_pending<T, MC, W, PSPEC>(col, pe, gv, xv, a, b);
    if (PSPEC || pe.on) {
      double yn[W], sn[W];
      newest_cols<MC, W, PSPEC>(col, a, b, yn, sn);
      const int64_t iw_ = tr.wrow(i);  // (the committed pair goes to the row's slot of the layout)
      if (NT) {
        stnt<W>(cwy + iw_, yn);
        stnt<W>(cws + iw_, sn);
      } else {
        st<W>(cwy + iw_, yn);
        st<W>(cws + iw_, sn);
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
        const double dk = subsm_dir<MC, PSPEC>(xv[k], zv[k], gv[k], ak, bk, col, theta, rtheta, cf, wv);
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
    // zout == dvec == nullptr ("lean"): the first trial step is 1, so x = z is stored below and
    // both z and d = x - t stay implicit until something needs them as vectors (solver.hip,
    // ensure_d) -- 5 store streams instead of 7.
    // tvec == rout == nullptr (ping-pong iterate buffers, lbfgsb_hip_setulb_dev_pp): t = x and
    // r = g are not copies but a change of roles -- the trial point goes to the OTHER x buffer
    // (xout; it holds t of the previous line search, which pd reads above, row by row before this
    // store), the caller's next gradient to the other g buffer: 3 store streams instead of 5.
    if (NT) {
      if (zout) stnt<W>(zout + i, zv);
      if (dvec) stnt<W>(dvec + i, dv);
      if (tvec) stnt<W>(tvec + i, xv);
      if (rout) stnt<W>(rout + i, gv);
      if (xout) stnt<W>(xout + i, zv);
    } else {
      if (zout) st<W>(zout + i, zv);
      if (dvec) st<W>(dvec + i, dv);
      if (tvec) st<W>(tvec + i, xv);  // t = x (:2235)
      if (rout) st<W>(rout + i, gv);  // r = g (:2236)
      // first trial point of the line search when its step is known to be 1: x = z (:2265);
      // xout may alias xx (each row is read above before it is written here)
      if (xout) st<W>(xout + i, zv);
    }
  };
  if constexpr (CW)
    for_tiles_cw<SubsmTripCW<T, MC, NT, PSPEC>>(n, ctx, lmask, body);
  else
    for_rows_raw<SubsmTrip<T, MC, V, NT, PSPEC>, SubsmTrip<T, MC, 1, NT, PSPEC>, V, PIPE, NS>(n, ctx, body);
  block_reduce_store<4>(acc, 3, 1, 0, part, pstride);
}
template <typename T>
void launch_subsm_update(Queue &q, int64_t n, double tsum, T *zout, const T *pr, T *rout, const T *l,
                         const T *u, const nb_t *nbd, const iw_t *iwhere, const T *xx, const T *gg,
                         WStore<T> w, int head, int col, double theta, const Coef &cf,
                         const Coef &wv, T *dvec, T *tvec, T *xout, int do_stpmx, Pend pe,
                         const T *pd, int ub) {
  int gr = 0;
  const int64_t slot = (int64_t)((head - 1 + col - 1) % w.m) * w.ld;  // physical column of col-1
  const bool spec = pe.on && col == maxc_for(col);
  double *part = q.part();
  const int pstride = MAX_BLOCKS;
  if (w.lmask) {  // W in the tile-local free-row layout
    bool done = false;
    if constexpr (sizeof(T) == 8) {
      if (col <= 10) {
#define LB_SUBSM_CW(MCV, NTV, PSPECV)                                                                           \
  {                                                                                                             \
    gr = grid_for_w(q, n, VecOf<T>::V, (const void *)&subsm_update_kernel<T, MCV, NTV, PSPECV, false, true>);   \
    hipLaunchKernelGGL((subsm_update_kernel<T, MCV, NTV, PSPECV, false, true>), dim3(gr), dim3(BLOCK), 0,       \
                       q.stream, n, tsum, zout, pr, rout, l, u, nbd, iwhere, xx, gg, w.ws, w.wy, w.zero, w.ld,   \
                       w.m, head, col, theta, cf, wv, dvec, tvec, xout, do_stpmx, pe, pd, w.wy + slot,          \
                       w.ws + slot, ub, part, pstride, w.lmask);                                                \
  }
        if (col <= 5) {
          if (q.nt) { if (spec) LB_SUBSM_CW(5, true, true) else LB_SUBSM_CW(5, true, false) }
          else { if (spec) LB_SUBSM_CW(5, false, true) else LB_SUBSM_CW(5, false, false) }
        } else {
          if (q.nt) { if (spec) LB_SUBSM_CW(10, true, true) else LB_SUBSM_CW(10, true, false) }
          else { if (spec) LB_SUBSM_CW(10, false, true) else LB_SUBSM_CW(10, false, false) }
        }
#undef LB_SUBSM_CW
        LB_LAUNCHED(q);
        finalize_from(q, part, pstride, gr, 3, 1, 0);
        done = true;
      }
    }
    if (!done && q.launch_err == hipSuccess)
      q.launch_err = hipErrorInvalidValue, q.launch_err_where = "subsm_update: compact W needs fp64, col <= 10";
    return;
  }
#define LB_SUBSM(PSPECV)                                                                            \
  DISPATCH_MAXC_NT(col, q.nt, DISPATCH_PIPE(MC, {                                                   \
                     gr = grid_for_w(q, n, VecOf<T>::V,                                             \
                                     (const void *)&subsm_update_kernel<T, MC, NTV, PSPECV, PIPEV>); \
                     hipLaunchKernelGGL((subsm_update_kernel<T, MC, NTV, PSPECV, PIPEV>), dim3(gr), \
                                        dim3(BLOCK), 0, q.stream, n, tsum, zout, pr, rout, l, u,    \
                                        nbd, iwhere, xx, gg, w.ws, w.wy, w.zero, w.ld, w.m, head,   \
                                        col, theta, cf, wv, dvec, tvec, xout, do_stpmx, pe, pd,     \
                                        w.wy + slot, w.ws + slot, ub, part, pstride);               \
                   }))
  if (spec)
    LB_SUBSM(true);
  else
    LB_SUBSM(false);
#undef LB_SUBSM
  LB_LAUNCHED(q);
  finalize_from(q, part, pstride, gr, 3, 1, 0);
}

// The Newton direction as a vector (free rows; 0 elsewhere), for the backtracking branch only.
template <typename T, int MC, bool NT>
__global__ __launch_bounds__(BLOCK) void subsm_dir_kernel(
    int64_t n, const T *__restrict__ xcp, const iw_t *__restrict__ iwhere,
    const T *__restrict__ xx, const T *__restrict__ gg, const T *__restrict__ ws,
    const T *__restrict__ wy, const T *__restrict__ zero, int64_t ldw, int m, int head, int col,
    double theta, Coef cf, Coef wv, T *__restrict__ ndir) {
  const double rtheta = 1.0 / theta;
  for_rows<T, RowsPer<T, MC>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double zv[W], xv[W], gv[W], a[MC][W], b[MC][W], out[W];
    int iw[W];
    ldx<W, NT>(xcp + i, zv);
    ldx<W, NT>(xx + i, xv);
    ldx<W, NT>(gg + i, gv);
    ldi<W>(iwhere + i, iw);
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, col, head, m, ldw) + i;
      ld_col<T, W, NT>(j < col, wy + off, zero, a[j]);
      ld_col<T, W, NT>(j < col, ws + off, zero, b[j]);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      out[k] = 0.0;
      if (iw[k] <= 0) {
        double ak[MC], bk[MC];
#pragma unroll
        for (int j = 0; j < MC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
        out[k] = subsm_dir<MC>(xv[k], zv[k], gv[k], ak, bk, col, theta, rtheta, cf, wv);
      }
    }
    st<W>(ndir + i, out);
  });
}
template <typename T>
void launch_subsm_dir(Queue &q, int64_t n, const T *xcp, const iw_t *iwhere, const T *xx,
                      const T *gg, WStore<T> w, int head, int col, double theta, const Coef &cf,
                      const Coef &wv, T *ndir) {
  const int gr = grid_for_w(q, n, VecOf<T>::V);
  DISPATCH_MAXC_NT(col, q.nt, hipLaunchKernelGGL((subsm_dir_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                        q.stream, n, xcp, iwhere, xx, gg, w.ws, w.wy, w.zero, w.ld, w.m,
                                        head, col, theta, cf, wv, ndir));
  LB_LAUNCHED(q);
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_subsm_update<T>(Queue &, int64_t, double, T *, const T *, T *, const T *, const T *, const nb_t *, const iw_t *, const T *, const T *, WStore<T>, int, int, double, const Coef &, const Coef &, T *, T *, T *, int, Pend, const T *, int); \
  template void launch_subsm_dir<T>(Queue &, int64_t, const T *, const iw_t *, const T *, const T *, WStore<T>, int, int, double, const Coef &, const Coef &, T *);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
