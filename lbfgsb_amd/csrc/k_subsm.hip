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
template <typename T, int MC, int W>
struct SubsmRegs {  // the register images of one row group
  RawOf<T, W> rl, ru, rx, rg, ra[MC], rb[MC];
  RawOf<nb_t, W> rnb;
  RawOf<iw_t, W> riw;
};
template <typename T, int MC, int W, bool NT, bool PSPEC>
struct SubsmTrip : SubsmRegs<T, MC, W>, NaturalRows<W> {
  static constexpr int NL = 6 + 2 * MC;
  __device__ __forceinline__ void issue(const SubsmCtx<T> &c, int64_t i) {
    constexpr int B = (int)sizeof(T) * W;
    raw_issue<B, NT>(this->rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(this->ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(this->rx, c.xx + i);
    raw_issue<B, NT>(this->rg, c.gg + i);
    raw_issue<W, false>(this->rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<W, false>(this->riw, c.iwhere + i);
    // a pending pair is read from (r, d) -- or (r, t) when d is implicit -- which this pass
    // overwrites further down
    issue_cols<T, MC, W, NT, PSPEC>(c.wy, c.ws, c.r, c.pd, c.zero, i, c.col, c.head, c.m, c.ldw, c.pe,
                                    this->ra, this->rb);
  }
  __device__ __forceinline__ void land() {}
};
// The tile-local free-row layout of W (for_tiles_cw): x, g, the bounds, iwhere and a pending pair's vectors from
// the row itself, the stored columns from the row's slot if its layout bit is set -- only free rows (iwhere <= 0)
// use them (cmprlb :1565-1583 and subsm :2770-2778 run over Index(1:nfree)); a free row whose bit is clear fetches
// them in the kernel body (reload_cols).
// where logical column j of a row comes from: the stored column at the row's slot, or -- the pending pair -- the
// vectors it is formed from at the row itself
template <typename T, int MC, bool PSPEC>
__device__ __forceinline__ void subsm_cw_col(const SubsmCtx<T> &c, int j, int64_t i, int64_t slot, bool want,
                                             const T *&py, const T *&ps) {
  if constexpr (PSPEC) {
    if (j == MC - 1) {
      py = c.r + i, ps = c.pd + i;
    } else {
      const int64_t off = (int64_t)((c.head - 1 + j) % c.m) * c.ldw + slot;
      py = want ? c.wy + off : c.zero, ps = want ? c.ws + off : c.zero;
    }
  } else {
    const int64_t off = col_off(j, c.col, c.head, c.m, c.ldw) + slot;
    const bool live = j < c.col, pj = c.pe.on && j == c.col - 1;
    py = !live ? c.zero : (pj ? c.r + i : (want ? c.wy + off : c.zero));
    ps = !live ? c.zero : (pj ? c.pd + i : (want ? c.ws + off : c.zero));
  }
}
template <typename T, int MC, bool NT, bool PSPEC>
struct SubsmTripCW1 : SubsmRegs<T, MC, 1>, CwOneRow {
  static constexpr int NL = 6 + 2 * MC;
  template <bool NTL>
  __device__ __forceinline__ void cols(const SubsmCtx<T> &c, int64_t srow) {
    constexpr int B = (int)sizeof(T);
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const T *py, *ps;
      subsm_cw_col<T, MC, PSPEC>(c, j, this->ri_, srow, true, py, ps);
      raw_issue<B, NTL>(this->ra[j], py);
      raw_issue<B, NTL>(this->rb[j], ps);
    }
  }
  // (`first`: a row whose layout bit is clear reads the first entry of its tile, see UpdScanTripCW1)
  __device__ __forceinline__ void issue_cw(const SubsmCtx<T> &c, int64_t i, int64_t slot, bool lf, int64_t first) {
    constexpr int B = (int)sizeof(T);
    this->ri_ = i, this->ws_ = slot, this->lf_ = lf;
    raw_issue<B, NT>(this->rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(this->ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(this->rx, c.xx + i);
    raw_issue<B, NT>(this->rg, c.gg + i);
    raw_issue<1, false>(this->rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<1, false>(this->riw, c.iwhere + i);
    cols<NT>(c, lf ? slot : first);
  }
  __device__ __forceinline__ void land() {}
};
// ... the rows lane, lane + 64 of a full tile as one row group of width 2 (see UpdScanTripCW2, k_update.hip)
template <typename T, int MC, bool NT, bool PSPEC>
struct SubsmTripCW2 : SubsmRegs<T, MC, 2>, CwPairRows {
  static_assert(sizeof(T) == 8, "compact W: fp64");
  static constexpr int NL = 2 * (6 + 2 * MC);
  RawReg<1> nb_[2], iw_[2];
  // the stored columns of row k from its slot (the pending pair's vectors are loaded with the row's other vectors)
  template <bool NTL>
  __device__ __forceinline__ void cols_row(const SubsmCtx<T> &c, int k, int64_t srow) {
#pragma unroll
    for (int j = 0; j < (PSPEC ? MC - 1 : MC); ++j) {
      const int64_t off = (PSPEC ? (int64_t)((c.head - 1 + j) % c.m) * c.ldw : col_off(j, c.col, c.head, c.m, c.ldw)) + srow;
      const bool stored = PSPEC || (j < c.col && !(c.pe.on && j == c.col - 1));
      if (PSPEC || stored || j >= c.col) {  // (not the pending column: it has been loaded from r / pd)
        raw_issue_half<NTL>(this->ra[j], k, stored ? c.wy + off : c.zero);
        raw_issue_half<NTL>(this->rb[j], k, stored ? c.ws + off : c.zero);
      }
    }
  }
  __device__ __forceinline__ void issue_cw(const SubsmCtx<T> &c, const CwTile &t) {
    this->tile_ = t;
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t i = t.tb + (lane + 64 * k);
      raw_issue_half<NT>(this->rl, k, (c.ub & 1) ? c.l : c.l + i);
      raw_issue_half<NT>(this->ru, k, (c.ub & 2) ? c.u : c.u + i);
      raw_issue_half<NT>(this->rx, k, c.xx + i);
      raw_issue_half<NT>(this->rg, k, c.gg + i);
      raw_issue<1, false>(nb_[k], (c.ub & 4) ? c.nbd : c.nbd + i);
      raw_issue<1, false>(iw_[k], c.iwhere + i);
      // the pending pair's y and s are formed from r and d / t of EVERY row (the pass commits the whole column)
      if constexpr (PSPEC) {
        raw_issue_half<NT>(this->ra[MC - 1], k, c.r + i);
        raw_issue_half<NT>(this->rb[MC - 1], k, c.pd + i);
      } else if (c.pe.on) {
#pragma unroll
        for (int j = 0; j < MC; ++j)
          if (j == c.col - 1) {
            raw_issue_half<NT>(this->ra[j], k, c.r + i);
            raw_issue_half<NT>(this->rb[j], k, c.pd + i);
          }
      }
    }
    int sl[2];
    bool lf[2];
    cw_slots(t, sl, lf);
    // (a row whose layout bit is clear reads the tile's first entry: see UpdScanTripCW2.  Plain loads for the W
    //  entries whatever NT says for the row vectors: the two runs of a tile -- rows l and rows l + 64 -- meet in one
    //  line, and a nontemporal line is fetched from HBM for each of the two instructions: 15.33 -> 13.51 GB per
    //  launch = 1.05 x the algorithmic bytes, 2.84 -> 2.71 ms at n = 1e8, same box)
    cols_row<false>(c, 0, t.tb + (lf[0] ? sl[0] : 0));
    cols_row<false>(c, 1, t.tb + (lf[1] ? sl[1] : 0));
  }
  __device__ __forceinline__ void land() {
    raw_join_bytes(this->rnb, nb_[0], nb_[1]);
    raw_join_bytes(this->riw, iw_[0], iw_[1]);
  }
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
  static_assert(!CW || (sizeof(T) == 8 && MC <= 10 && !PIPE), "compact W: fp64, MC <= 10, one trip in flight");
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
      // A free row whose layout bit is clear cannot exist HERE: the solver re-sorts the tiles in front of this pass
      // whenever a row changed status (cw_maybe_pack), so the bits are the free set of this very iteration -- or all
      // ones.  (Fetching such a row's entries in this kernel, as the update pass does, costs 88 registers and the
      // third wave per SIMD.)  Should it happen all the same, the count of bound hits becomes absurd and the host
      // refuses the result (subspace_land).
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (!tr.lf(k) && iw[k] <= 0) acc[0] += 1.0e30;
    }
    get_cols<T, MC, W>(tr.ra, tr.rb, a, b);
    fix_pending<T, MC, W, PSPEC>(col, pe, gv, xv, a, b);
    if (PSPEC || pe.on) {
      double yn[W], sn[W];
      newest_cols<MC, W, PSPEC>(col, a, b, yn, sn);
      // (the committed pair goes to the rows' slots of the layout)
      tr.template st_w<NT>(cwy, i, yn);
      tr.template st_w<NT>(cws, i, sn);
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
    if (zout) tr.template st_rows<NT>(zout, i, zv);
    if (dvec) tr.template st_rows<NT>(dvec, i, dv);
    if (tvec) tr.template st_rows<NT>(tvec, i, xv);  // t = x (:2235)
    if (rout) tr.template st_rows<NT>(rout, i, gv);  // r = g (:2236)
    // first trial point of the line search when its step is known to be 1: x = z (:2265);
    // xout may alias xx (each row is read above before it is written here)
    if (xout) tr.template st_rows<NT>(xout, i, zv);
  };
  if constexpr (CW)
    for_tiles_cw<SubsmTripCW2<T, MC, NT, PSPEC>, SubsmTripCW1<T, MC, NT, PSPEC>>(n, ctx, lmask, body);
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
