// k_wide.hip -- m > 32: the same iteration out of UNFUSED pieces (part of the gfx950 kernel set).
//
// The fused passes of k_update / k_subsm / k_cmprlb keep all 2 col operands of a row group in
// registers and are unrolled for at most MAXM = 32 pairs.  The reference puts no limit on m
// (src/lbfgsb.f90:93-97); beyond 32 the solver composes the iteration from the two tile primitives
// below plus the existing element-wise kernels, in tiles of <= 32 logical columns (circular
// addressing makes a tile just another (head, col) pair):
//   W' v            launch_wtv on a tile                      (cauchy's p, subsm's wv, matupd's dots,
//                                                              formk's inner products with masked columns)
//   out += W c      tile_axpy_kernel                          (cmprlb's r, subsm's Newton direction)
// Memory-bound and simple on purpose: this is the completeness path, not the fast one (m <= 32 never
// comes here).  Element-wise arithmetic follows the reference's operation order like everywhere else.
#include "kernels_common.hpp"

namespace lbk {

// out_i (+)= sum_j (Wy(i,j) * a_j) / div + Ws(i,j) * b_j over the tc <= 32 logical columns of one tile,
// j ascending as cmprlb :1576-1581 and subsm :2770-2778 run; rows with mask (iwhere > 0: not free) are
// left alone when masked != 0.  div = theta for subsm's first term (the reference divides the PRODUCT).
// 16 bytes per lane and column (2 rows in fp64, 4 in fp32), the columns in groups of 8 whose 16 loads are issued
// together (slots beyond tc read the zero buffer: no branch between the loads); the sum of a row runs over j in
// the same order as before, term by term.  (Round 4's form -- one row per lane, a run-time loop over the columns
// -- moved 5.6 TB/s at m = 48; this is the second pass over W of every m > 32 iteration.)
template <typename T, bool NT>
__global__ __launch_bounds__(BLOCK) void tile_axpy_kernel(int64_t n, const T *__restrict__ ws,
                                                          const T *__restrict__ wy, const T *__restrict__ zero,
                                                          int64_t ldw, int m, int head, int tc, Coef cf, double div,
                                                          const iw_t *__restrict__ iwhere, int masked, T *out) {
  constexpr int G = 8;
  for_rows<T, VecOf<T>::V>(n, [&](int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double acc[W], old[W];
    int iw[W];
    ld<W>(out + i, acc);
#pragma unroll
    for (int k = 0; k < W; ++k) old[k] = acc[k];
    if (masked) {
      ldi<W>(iwhere + i, iw);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) iw[k] = 0;
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < W; ++k) any = any || iw[k] <= 0;
    if (__ballot(any) == 0ull) return;  // (a wave whose rows are all masked reads nothing)
    for (int j0 = 0; j0 < tc; j0 += G) {
      double a[G][W], b[G][W];
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        const int j = j0 + jj;
        const int64_t off = (int64_t)((head - 1 + (j < tc ? j : 0)) % m) * ldw + i;
        ld_col<T, W, NT>(j < tc, wy + off, zero, a[jj]);
        ld_col<T, W, NT>(j < tc, ws + off, zero, b[jj]);
      }
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        if (j0 + jj < tc) {  // (uniform)
          const double ca = cf.a[j0 + jj], cb = cf.a[MAXM + j0 + jj];
#pragma unroll
          for (int k = 0; k < W; ++k) acc[k] = acc[k] + (a[jj][k] * ca) / div + b[jj][k] * cb;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = iw[k] > 0 ? old[k] : acc[k];  // (rows that are not free keep their value)
    st<W>(out + i, acc);
  });
}
template <typename T>
void launch_tile_axpy(Queue &q, int64_t n, WStore<T> w, int head, int tc, const Coef &cf, double div,
                      const iw_t *iwhere, int masked, T *out) {
  const int gr = grid_for(n, VecOf<T>::V);
  if (q.nt)
    hipLaunchKernelGGL((tile_axpy_kernel<T, true>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws, w.wy, w.zero, w.ld,
                       w.m, head, tc, cf, div, iwhere, masked, out);
  else
    hipLaunchKernelGGL((tile_axpy_kernel<T, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws, w.wy, w.zero, w.ld,
                       w.m, head, tc, cf, div, iwhere, masked, out);
  LB_LAUNCHED(q);
}

// The r pass with cmprlb's start and subsm's tail folded into its first / last tile (kernels.hpp, WideTail).  Per
// row the arithmetic -- and, in REAL32, every rounding to T the unfused kernels put between their steps -- is
// that of cmprlb_init_kernel, tile_axpy_kernel, subsm_project_kernel and lnsrlb_begin_kernel run one after the
// other.
template <typename T, bool NT, bool FIRST, bool LAST>
__global__ __launch_bounds__(BLOCK) void tile_axpy_fused_kernel(int64_t n, const T *__restrict__ ws,
                                                                const T *__restrict__ wy, const T *__restrict__ zero,
                                                                int64_t ldw, int m, int head, int tc, Coef cf,
                                                                const iw_t *__restrict__ iwhere, T *out, WideTail<T> wt,
                                                                double *part) {
  constexpr int G = 8;
  double red[4] = {0.0, 0.0, 0.0, 1.0e10};
  const double rtheta = 1.0 / wt.theta;
  for_rows<T, VecOf<T>::V>(n, [&](int64_t i, auto wtag) {
    constexpr int W = decltype(wtag)::value;
    double acc[W], xv[W], gv[W], lv[W], uv[W], zc[W];
    int iw[W];
    ldi<W>(iwhere + i, iw);
    if constexpr (FIRST || LAST) {
      ld<W>(wt.x + i, xv);
      ld<W>(wt.g + i, gv);
      ld<W>(wt.l + i, lv);
      ld<W>(wt.u + i, uv);
#pragma unroll
      for (int k = 0; k < W; ++k) zc[k] = (double)(T)xcp_row<T>(xv[k], gv[k], iw[k], lv[k], uv[k], wt.tsum);
    }
    if constexpr (FIRST) {
#pragma unroll
      for (int k = 0; k < W; ++k)  // cmprlb_init_kernel
        acc[k] = iw[k] > 0 ? 0.0 : (double)(T)(wt.plain ? -gv[k] : -wt.theta * (zc[k] - xv[k]) - gv[k]);
    } else {
      ld<W>(out + i, acc);
    }
    for (int j0 = 0; j0 < tc; j0 += G) {
      double a[G][W], b[G][W];
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        const int j = j0 + jj;
        const int64_t off = (int64_t)((head - 1 + (j < tc ? j : 0)) % m) * ldw + i;
        ld_col<T, W, NT>(j < tc, wy + off, zero, a[jj]);
        ld_col<T, W, NT>(j < tc, ws + off, zero, b[jj]);
      }
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        if (j0 + jj < tc) {  // (uniform)
          const double ca = cf.a[j0 + jj], cb = cf.a[MAXM + j0 + jj];
#pragma unroll
          for (int k = 0; k < W; ++k)
            if (iw[k] <= 0) acc[k] = acc[k] + a[jj][k] * ca + b[jj][k] * cb;  // (tile_axpy_kernel with div = 1)
        }
      }
    }
    if constexpr (!LAST) {
      st<W>(out + i, acc);
    } else {
      double zv[W], dv[W];
      int nbk[W];
      ldi<W>(wt.nbd + i, nbk);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        zv[k] = zc[k];
        if (iw[k] <= 0) {  // subsm_project_kernel
          const double dk = rtheta * (double)(T)acc[k];
          const double xk = zc[k];
          if (nbk[k] != 0) {
            if (nbk[k] == 1) {
              zv[k] = fmax(lv[k], xk + dk);
              if (zv[k] == lv[k]) red[0] += 1.0;
            } else if (nbk[k] == 2) {
              const double t1 = fmax(lv[k], xk + dk);
              zv[k] = fmin(uv[k], t1);
              if (zv[k] == lv[k] || zv[k] == uv[k]) red[0] += 1.0;
            } else if (nbk[k] == 3) {
              zv[k] = fmin(uv[k], xk + dk);
              if (zv[k] == uv[k]) red[0] += 1.0;
            }
          } else {
            zv[k] = xk + dk;
          }
          zv[k] = (double)(T)zv[k];
        }
        dv[k] = zv[k] - xv[k];              // mainlb :720-722 (lnsrlb_begin_kernel)
        red[1] = red[1] + dv[k] * gv[k];    // dd_p (:2824-2827) == g'd (:2244)
        red[2] = red[2] + dv[k] * dv[k];    // dtd (:2196)
        if (wt.do_stpmx && nbk[k] != 0) {   // :2206-2225
          const double a1 = dv[k];
          if (a1 < 0.0 && nbk[k] <= 2) {
            const double a2 = lv[k] - xv[k];
            red[3] = fmin(red[3], a2 >= 0.0 ? 0.0 : a2 / a1);
          } else if (a1 > 0.0 && nbk[k] >= 2) {
            const double a2 = uv[k] - xv[k];
            red[3] = fmin(red[3], a2 <= 0.0 ? 0.0 : a2 / a1);
          }
        }
      }
      if (wt.zout) st<W>(wt.zout + i, zv);
      if (wt.dvec) st<W>(wt.dvec + i, dv);
      if (wt.tvec) st<W>(wt.tvec + i, xv);  // t = x (:2235)
      if (wt.rout) st<W>(wt.rout + i, gv);  // r = g (:2236)
      if (wt.xout) st<W>(wt.xout + i, zv);  // the first trial point x = z (:2265); may alias wt.x (row read above)
    }
  });
  if constexpr (LAST) block_reduce_store<4>(red, 3, 1, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_tile_axpy_fused(Queue &q, int64_t n, WStore<T> w, int head, int tc, const Coef &cf, const iw_t *iwhere,
                            T *out, int first, int last, const WideTail<T> &wt) {
  const int gr = grid_for(n, VecOf<T>::V);
#define LB_TAF(NTV, FV, LV)                                                                                         \
  hipLaunchKernelGGL((tile_axpy_fused_kernel<T, NTV, FV, LV>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws, w.wy,   \
                     w.zero, w.ld, w.m, head, tc, cf, iwhere, out, wt, q.part())
  if (q.nt) {
    if (first && last) LB_TAF(true, true, true);
    else if (first) LB_TAF(true, true, false);
    else if (last) LB_TAF(true, false, true);
    else LB_TAF(true, false, false);
  } else {
    if (first && last) LB_TAF(false, true, true);
    else if (first) LB_TAF(false, true, false);
    else if (last) LB_TAF(false, false, true);
    else LB_TAF(false, false, false);
  }
#undef LB_TAF
  LB_LAUNCHED(q);
  if (last) launch_finalize(q, gr, 3, 1, 0);
}

// The r pass in one launch (kernels.hpp, launch_wide_r_pass): tile_axpy_fused_kernel<FIRST, LAST> over all col
// columns.  Per row the arithmetic is that of the tiled launches -- in REAL32 including the rounding to T their
// partial sum takes when it goes through memory after every MAXM columns -- with the newest pair taken from
// (r, d) in its stored form (pend_y / pend_sx) and committed.
template <typename T, bool NT>
__global__ __launch_bounds__(BLOCK) void wide_r_pass_kernel(int64_t n, const T *__restrict__ ws,
                                                            const T *__restrict__ wy, const T *__restrict__ zero,
                                                            int64_t ldw, int m, int head, int col, CoefWide cf,
                                                            const iw_t *__restrict__ iwhere,
                                                            const nb_t *__restrict__ nbd8, int ub, WideTail<T> wt,
                                                            Pend pe, const T *pr, const T *pd, T *cwy, T *cws,
                                                            double *part) {
  constexpr int G = 8;
  double red[4] = {0.0, 0.0, 0.0, 1.0e10};
  const double rtheta = 1.0 / wt.theta;
  const int jp = pe.on ? col - 1 : -1;  // the pending column
  __shared__ T dict[16];
  dict_fill<T>(dict, wt.l, wt.u, ub);
  for_rows<T, VecOf<T>::V>(n, [&](int64_t i, auto wtag) {
    constexpr int W = decltype(wtag)::value;
    double acc[W], xv[W], gv[W], lv[W], uv[W], zc[W];
    int iw[W], nbk[W];
    ldi<W>(iwhere + i, iw);
    ld<W>(wt.x + i, xv);
    ld<W>(wt.g + i, gv);
    ld<W>((ub & 1) ? wt.l : wt.l + i, lv);
    ld<W>((ub & 2) ? wt.u : wt.u + i, uv);
    ldi<W>((ub & 4) ? nbd8 : nbd8 + i, nbk);
    dict_apply<T, W>(dict, ub, nbk, lv, uv);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      zc[k] = (double)(T)xcp_row<T>(xv[k], gv[k], iw[k], lv[k], uv[k], wt.tsum);
      acc[k] = iw[k] > 0 ? 0.0 : (double)(T)(wt.plain ? -gv[k] : -wt.theta * (zc[k] - xv[k]) - gv[k]);  // cmprlb_init_kernel
    }
    for (int j0 = 0; j0 < col; j0 += G) {
      double a[G][W], b[G][W];
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        const int j = j0 + jj;
        const int64_t off = (int64_t)((head - 1 + (j < col ? j : 0)) % m) * ldw + i;
        const T *py = j == jp ? pr + i : wy + off, *ps = j == jp ? pd + i : ws + off;
        ld_col<T, W, NT>(j < col, py, zero, a[jj]);
        ld_col<T, W, NT>(j < col, ps, zero, b[jj]);
      }
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        const int j = j0 + jj;
        if (j < col) {  // (uniform)
          if (j == jp) {
#pragma unroll
            for (int k = 0; k < W; ++k) {
              a[jj][k] = pend_y<T>(gv[k], a[jj][k]);
              b[jj][k] = pend_sx<T>(b[jj][k], xv[k], pe);
            }
            st<W>(cwy + i, a[jj]);
            st<W>(cws + i, b[jj]);
          }
          const double ca = cf.a[j], cb = cf.a[WIDE_MAXC + j];
#pragma unroll
          for (int k = 0; k < W; ++k)
            if (iw[k] <= 0) acc[k] = acc[k] + a[jj][k] * ca + b[jj][k] * cb;  // (tile_axpy_kernel with div = 1)
        }
      }
      if ((j0 + G) % MAXM == 0 && j0 + G < col) {  // where the tiled launches store r and load it again
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = (double)(T)acc[k];
      }
    }
    double zv[W], dv[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      zv[k] = zc[k];
      if (iw[k] <= 0) {  // subsm_project_kernel
        const double dk = rtheta * (double)(T)acc[k];
        const double xk = zc[k];
        if (nbk[k] != 0) {
          if (nbk[k] == 1) {
            zv[k] = fmax(lv[k], xk + dk);
            if (zv[k] == lv[k]) red[0] += 1.0;
          } else if (nbk[k] == 2) {
            const double t1 = fmax(lv[k], xk + dk);
            zv[k] = fmin(uv[k], t1);
            if (zv[k] == lv[k] || zv[k] == uv[k]) red[0] += 1.0;
          } else if (nbk[k] == 3) {
            zv[k] = fmin(uv[k], xk + dk);
            if (zv[k] == uv[k]) red[0] += 1.0;
          }
        } else {
          zv[k] = xk + dk;
        }
        zv[k] = (double)(T)zv[k];
      }
      dv[k] = zv[k] - xv[k];              // mainlb :720-722 (lnsrlb_begin_kernel)
      red[1] = red[1] + dv[k] * gv[k];    // dd_p (:2824-2827) == g'd (:2244)
      red[2] = red[2] + dv[k] * dv[k];    // dtd (:2196)
      if (wt.do_stpmx && nbk[k] != 0) {   // :2206-2225
        const double a1 = dv[k];
        if (a1 < 0.0 && nbk[k] <= 2) {
          const double a2 = lv[k] - xv[k];
          red[3] = fmin(red[3], a2 >= 0.0 ? 0.0 : a2 / a1);
        } else if (a1 > 0.0 && nbk[k] >= 2) {
          const double a2 = uv[k] - xv[k];
          red[3] = fmin(red[3], a2 <= 0.0 ? 0.0 : a2 / a1);
        }
      }
    }
    if (wt.zout) st<W>(wt.zout + i, zv);
    if (wt.dvec) st<W>(wt.dvec + i, dv);
    if (wt.tvec) st<W>(wt.tvec + i, xv);  // t = x (:2235)
    if (wt.rout) st<W>(wt.rout + i, gv);  // r = g (:2236)
    if (wt.xout) st<W>(wt.xout + i, zv);  // the first trial point x = z (:2265); may alias wt.x and pd (rows read above)
  });
  block_reduce_store<4>(red, 3, 1, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_wide_r_pass(Queue &q, int64_t n, WStore<T> w, int head, int col, const CoefWide &cf, const iw_t *iwhere,
                        const nb_t *nbd8, int ub, const WideTail<T> &wt, Pend pe, const T *pr, const T *pd) {
  const int gr = grid_for(n, VecOf<T>::V);
  const int64_t slot = (int64_t)((head - 1 + col - 1) % w.m) * w.ld;  // physical column of col - 1
  if (q.nt)
    hipLaunchKernelGGL((wide_r_pass_kernel<T, true>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws, w.wy, w.zero, w.ld,
                       w.m, head, col, cf, iwhere, nbd8, ub, wt, pe, pr, pd, w.wy + slot, w.ws + slot, q.part());
  else
    hipLaunchKernelGGL((wide_r_pass_kernel<T, false>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws, w.wy, w.zero, w.ld,
                       w.m, head, col, cf, iwhere, nbd8, ub, wt, pe, pr, pd, w.wy + slot, w.ws + slot, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 3, 1, 0);
}

// out_i = src_i on the rows selected (want_free: iwhere <= 0, else iwhere > 0), 0 elsewhere
template <typename T>
__global__ __launch_bounds__(BLOCK) void masked_copy_kernel(int64_t n, const T *__restrict__ src,
                                                            const iw_t *__restrict__ iwhere, int want_free,
                                                            T *__restrict__ out) {
  for_rows<T, 1>(n, [&](int64_t i, auto) {
    const bool fr = iwhere[i] <= 0;
    out[i] = (fr == (want_free != 0)) ? src[i] : (T)0;
  });
}
template <typename T>
void launch_masked_copy(Queue &q, int64_t n, const T *src, const iw_t *iwhere, int want_free, T *out) {
  hipLaunchKernelGGL(masked_copy_kernel<T>, dim3(grid_for(n, 1)), dim3(BLOCK), 0, q.stream, n, src, iwhere,
                     want_free, out);
  LB_LAUNCHED(q);
}

// the rows of a (sorted) changed-row list as dense records: out[k][c] = Wy(row_k, c) for c < upcl, Ws(row_k,
// c - upcl) beyond -- what formk's patch for entering / leaving variables (:1801-1851) needs, for any number of
// pairs (formk_patch_kernel keeps its tiles in LDS and is sized for 32)
template <typename T>
__global__ __launch_bounds__(BLOCK) void rows_gather_kernel(const uint32_t *__restrict__ chg, uint32_t cnt,
                                                            const T *__restrict__ ws, const T *__restrict__ wy,
                                                            int64_t ldw, int m, int head, int upcl,
                                                            double *__restrict__ out) {
  const int64_t total = (int64_t)cnt * 2 * upcl;
  for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
    const int64_t k = e / (2 * upcl);
    const int c = (int)(e % (2 * upcl)), jj = c < upcl ? c : c - upcl;
    const int64_t off = (int64_t)((head - 1 + jj) % m) * ldw + (int64_t)(chg[k] & 0x7FFFFFFFu);
    out[e] = c < upcl ? (double)wy[off] : (double)ws[off];
  }
}
template <typename T>
void launch_rows_gather(Queue &q, const uint32_t *chg, uint32_t cnt, WStore<T> w, int head, int upcl, double *out) {
  const int64_t total = (int64_t)cnt * 2 * upcl;
  int64_t gr = (total + BLOCK - 1) / BLOCK;
  if (gr < 1) gr = 1;
  if (gr > MAX_BLOCKS) gr = MAX_BLOCKS;
  hipLaunchKernelGGL(rows_gather_kernel<T>, dim3((int)gr), dim3(BLOCK), 0, q.stream, chg, cnt, w.ws, w.wy, w.ld, w.m,
                     head, upcl, out);
  LB_LAUNCHED(q);
}

// cauchy's direction as a vector (:1270-1330): d_i = -g_i for the variables that move (tbrk >= 0, incl.
// +inf), 0 for the others (tbrk = -1), from the breakpoint times the scan has just written
template <typename T>
__global__ __launch_bounds__(BLOCK) void cauchy_dvec_kernel(int64_t n, const T *__restrict__ g,
                                                            const T *__restrict__ tbrk, T *__restrict__ out) {
  for_rows<T, 1>(n, [&](int64_t i, auto) { out[i] = (double)tbrk[i] >= 0.0 ? (T)(-(double)g[i]) : (T)0; });
}
template <typename T>
void launch_cauchy_dvec(Queue &q, int64_t n, const T *g, const T *tbrk, T *out) {
  hipLaunchKernelGGL(cauchy_dvec_kernel<T>, dim3(grid_for(n, 1)), dim3(BLOCK), 0, q.stream, n, g, tbrk, out);
  LB_LAUNCHED(q);
}

// cmprlb's starting value (:1560-1567) on the free rows, 0 elsewhere: plain != 0 (unconstrained with pairs
// stored): r = -g; else r = -theta (z - x) - g with z the Cauchy point
template <typename T>
__global__ __launch_bounds__(BLOCK) void cmprlb_init_kernel(int64_t n, const T *__restrict__ x,
                                                            const T *__restrict__ g, const T *__restrict__ z,
                                                            const iw_t *__restrict__ iwhere, double theta,
                                                            int plain, T *__restrict__ out) {
  for_rows<T, 1>(n, [&](int64_t i, auto) {
    if (iwhere[i] > 0) {
      out[i] = (T)0;
    } else if (plain) {
      out[i] = (T)(-(double)g[i]);
    } else {
      out[i] = (T)(-theta * ((double)z[i] - (double)x[i]) - (double)g[i]);
    }
  });
}
template <typename T>
void launch_cmprlb_init(Queue &q, int64_t n, const T *x, const T *g, const T *z, const iw_t *iwhere,
                        double theta, int plain, T *out) {
  hipLaunchKernelGGL(cmprlb_init_kernel<T>, dim3(grid_for(n, 1)), dim3(BLOCK), 0, q.stream, n, x, g, z, iwhere,
                     theta, plain, out);
  LB_LAUNCHED(q);
}

// subsm's projected step (:2780-2827) from the Newton direction as a vector: dir_i *= 1/theta (dscal :2780,
// written back: the backtracking branch needs it), z_i = P(z_i + dir_i) on the free rows with the bound
// rules of :2789-2816.  res: sum [0] = #bound hits (iword), [1] = dd_p = sum (z_i - x_i) g_i over ALL rows
template <typename T>
__global__ __launch_bounds__(BLOCK) void subsm_project_kernel(int64_t n, T *z, T *dir,
                                                              const T *__restrict__ x, const T *__restrict__ g,
                                                              const T *__restrict__ l, const T *__restrict__ u,
                                                              const int32_t *__restrict__ nbd,
                                                              const iw_t *__restrict__ iwhere, double rtheta,
                                                              double *part) {
  double acc[2] = {0.0, 0.0};
  for_rows<T, 1>(n, [&](int64_t i, auto) {
    double zk = (double)z[i];
    if (iwhere[i] <= 0) {
      const double dk = rtheta * (double)dir[i];
      dir[i] = (T)dk;
      const double xk = zk, lk = (double)l[i], uk = (double)u[i];
      const int nb = nbd[i];
      if (nb != 0) {
        if (nb == 1) {
          zk = fmax(lk, xk + dk);
          if (zk == lk) acc[0] += 1.0;
        } else if (nb == 2) {
          const double t1 = fmax(lk, xk + dk);
          zk = fmin(uk, t1);
          if (zk == lk || zk == uk) acc[0] += 1.0;
        } else if (nb == 3) {
          zk = fmin(uk, xk + dk);
          if (zk == uk) acc[0] += 1.0;
        }
      } else {
        zk = xk + dk;
      }
      z[i] = (T)zk;
      zk = (double)(T)zk;
    }
    acc[1] = acc[1] + (zk - (double)x[i]) * (double)g[i];
  });
  block_reduce_store<2>(acc, 2, 0, 0, part, MAX_BLOCKS);
}
template <typename T>
void launch_subsm_project(Queue &q, int64_t n, T *z, T *dir, const T *x, const T *g, const T *l, const T *u,
                          const int32_t *nbd, const iw_t *iwhere, double rtheta) {
  const int gr = grid_for(n, 1);
  hipLaunchKernelGGL(subsm_project_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, n, z, dir, x, g, l, u, nbd,
                     iwhere, rtheta, q.part());
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 2, 0, 0);
}

#define INSTANTIATE(T) \
  template void launch_tile_axpy<T>(Queue &, int64_t, WStore<T>, int, int, const Coef &, double, const iw_t *, int, T *); \
  template void launch_tile_axpy_fused<T>(Queue &, int64_t, WStore<T>, int, int, const Coef &, const iw_t *, T *, int, int, const WideTail<T> &); \
  template void launch_wide_r_pass<T>(Queue &, int64_t, WStore<T>, int, int, const CoefWide &, const iw_t *, const nb_t *, int, const WideTail<T> &, Pend, const T *, const T *); \
  template void launch_masked_copy<T>(Queue &, int64_t, const T *, const iw_t *, int, T *); \
  template void launch_rows_gather<T>(Queue &, const uint32_t *, uint32_t, WStore<T>, int, int, double *); \
  template void launch_cauchy_dvec<T>(Queue &, int64_t, const T *, const T *, T *); \
  template void launch_cmprlb_init<T>(Queue &, int64_t, const T *, const T *, const T *, const iw_t *, double, int, T *); \
  template void launch_subsm_project<T>(Queue &, int64_t, T *, T *, const T *, const T *, const T *, const T *, const int32_t *, const iw_t *, double);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
