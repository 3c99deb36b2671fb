// k_subsm.hip -- subsm: Newton direction, projected step, line-search set-up, pair commit
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

namespace lbk {

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

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_subsm_update<T>(Queue &, int64_t, double, T *, T *, const T *, const T *, const int32_t *, const iw_t *, const T *, const T *, WStore<T>, int, int, double, const Coef &, int, const Coef &, T *, T *, T *, int, Pend); \
  template void launch_subsm_dir<T>(Queue &, int64_t, const T *, const iw_t *, const T *, const T *, WStore<T>, int, int, double, const Coef &, int, const Coef &, T *);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
