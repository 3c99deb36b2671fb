// k_cmprlb.hip -- cmprlb fused with W'r of subsm and formk's new row
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

namespace lbk {

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
// PSPEC: pe.on && col == MC (steady state once the memory is full): no per-column selects.
// PIPE: two trips in flight per wave (device_util.hpp, for_rows_raw) -- for the instantiations
// that hold more than 256 registers and run one wave per SIMD.
template <typename T>
struct CmprlbCtx {
  const T *x, *g, *ws, *wy, *zero, *pr, *pd;
  const iw_t *iwhere;
  int64_t ldw;
  int m, head, col;
  Pend pe;
};
template <typename T, int MC, int W, bool NT, bool PSPEC>
struct CmprlbTrip {
  static constexpr int NL = 3 + 2 * MC;
  RawOf<T, W> rg, rx, ra[MC], rb[MC];
  RawOf<iw_t, W> riw;
  __device__ __forceinline__ void issue(const CmprlbCtx<T> &c, int64_t i) {
    constexpr int B = (int)sizeof(T) * W;
    raw_issue<B, NT>(rg, c.g + i);
    raw_issue<B, NT>(rx, c.x + i);
    raw_issue<W, false>(riw, c.iwhere + i);
    issue_cols<T, MC, W, NT, PSPEC>(c.wy, c.ws, c.pr, c.pd, c.zero, i, c.col, c.head, c.m, c.ldw, c.pe, ra,
                                    rb);
  }
  __device__ __forceinline__ void land() {
    raw_land(rg);
    raw_land(rx);
    raw_land(riw);
    land_cols<T, MC, W>(ra, rb);
  }
};
template <typename T, int MC, bool NEWROW, bool NT, bool PSPEC, bool PIPE>
__global__ __launch_bounds__(BLOCK) void cmprlb_wtv_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ g, double tsum,
    const iw_t *__restrict__ iwhere, const T *__restrict__ ws, const T *__restrict__ wy,
    const T *__restrict__ zero, int64_t ldw, int m, int head, int col, double theta, Coef cf,
    const T *pr, const T *pd, Pend pe, double *part) {
  constexpr int NA = NEWROW ? 6 * MC : 2 * MC;
  constexpr int V = RowsPerAcc<T, MC, NA>::V;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const CmprlbCtx<T> ctx{x, g, ws, wy, zero, pr, pd, iwhere, ldw, m, head, col, pe};
  for_rows_raw<CmprlbTrip<T, MC, V, NT, PSPEC>, CmprlbTrip<T, MC, 1, NT, PSPEC>, V, PIPE, 0>(
      n, ctx, [&](auto &tr, int64_t, auto wt) {
    constexpr int W = decltype(wt)::value;
    // columns are read from the landed registers where they are used (col_pair): keeping all
    // 2*MC operands widened next to 6*MC fp64 sums does not fit the register file for fp32
    double xv[W], gv[W], rv[W];
    int iw[W];
    raw_get<W>(tr.rg, (const T *)nullptr, gv);
    raw_get<W>(tr.rx, (const T *)nullptr, xv);
    raw_geti<W>(tr.riw, (const iw_t *)nullptr, iw);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const double zk = xcp_free<T>(xv[k], gv[k], iw[k], tsum);  // only free rows are used
      rv[k] = -theta * (zk - xv[k]) - gv[k];
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      double aj[W], bj[W];
      col_pair<T, MC, W, PSPEC, false>(tr.ra, tr.rb, j, col, pe, gv, aj, bj);
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (PSPEC || j < col) rv[k] = rv[k] + aj[k] * cf.a[j] + bj[k] * cf.a[MAXM + j];
    }
#pragma unroll
    for (int k = 0; k < W; ++k) rv[k] = iw[k] <= 0 ? rv[k] : 0.0;
    double yf[W], sa[W];
    if constexpr (NEWROW) {
      double yn[W], sn[W];
      if constexpr (PSPEC) {
        col_pair<T, MC, W, PSPEC, false>(tr.ra, tr.rb, MC - 1, col, pe, gv, yn, sn);
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) yn[k] = 0.0, sn[k] = 0.0;
#pragma unroll
        for (int j = 0; j < MC; ++j) {
          double aj[W], bj[W];
          col_pair<T, MC, W, PSPEC, false>(tr.ra, tr.rb, j, col, pe, gv, aj, bj);
#pragma unroll
          for (int k = 0; k < W; ++k) {
            yn[k] = j == col - 1 ? aj[k] : yn[k];
            sn[k] = j == col - 1 ? bj[k] : sn[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < W; ++k) {
        yf[k] = iw[k] <= 0 ? yn[k] : 0.0;  // free rows
        sa[k] = iw[k] <= 0 ? 0.0 : sn[k];  // active rows
      }
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      double aj[W], bj[W];
      col_pair<T, MC, W, PSPEC, true>(tr.ra, tr.rb, j, col, pe, gv, aj, bj);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += aj[k] * rv[k];
        acc[MC + j] += bj[k] * rv[k];
        if constexpr (NEWROW) {
          acc[2 * MC + j] += yf[k] * aj[k];  // temp1 (:1764)
          acc[3 * MC + j] += sa[k] * bj[k];  // temp2 (:1769)
          acc[4 * MC + j] += sa[k] * aj[k];  // temp3 (:1770)
          acc[5 * MC + j] += bj[k] * yf[k];  // temp3 of the new column (:1789)
        }
      }
    }
  });
  block_reduce_store<NA>(acc, NA, 0, 0, part, MAX_BLOCKS);
}
// The same pass where 2 MC operand values + the accumulators of all MC columns do not fit the register
// file of one lane (MC = 20 with the new-row sums: 6 MC accumulators; MC = 32: 64 operands + 64 sums --
// the plain kernel spilled to scratch there, and a scratch reload inside the row loop returns in order
// behind the loads in flight).  Two neighbouring lanes share the work on their two row groups: every
// lane still loads all columns of its own rows (r needs them), but accumulates only ONE HALF of the
// columns -- for its own rows and, over DPP quad_perm moves, for its neighbour's -- so half the
// accumulators suffice, and the operands stay in storage precision until they are used.  Per-element
// arithmetic is unchanged; the sums are merely grouped differently.  Rows without a neighbour (odd group
// count, scalar tail) are loaded by both lanes 0 and 1 of the first workgroup, each taking its half.
// fp64, m = 20, n = 1e8: 7.5 -> 5.4 ms (4.5 -> 6.3 TB/s).  NEWROW = false: col 21..32 (formk runs from
// scratch there, no new-row sums), both types; NEWROW = true: MC = 20, fp64 always, fp32 when the
// pending-pair selects of the general shape would push the plain kernel into scratch.
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
// A value that is selected between two elements of a register array: pinned to a register first.
// Otherwise the compiler folds "c ? a[p] : a[q]" into ONE load through a selected address, which turns
// the whole array into a scratch allocation (the fp32 MC = 32 instantiation: 528 bytes per lane).
template <typename U>
__device__ __forceinline__ U in_reg(U v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <typename T, int MC, bool NT, bool NEWROW>
__global__ __launch_bounds__(BLOCK) void cmprlb_wtv_pair_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ g, double tsum,
    const iw_t *__restrict__ iwhere, const T *__restrict__ ws, const T *__restrict__ wy,
    int64_t ldw, int m, int head, int col, double theta, Coef cf, const T *pr,
    const T *pd, Pend pe, double *part) {
  constexpr int H = MC / 2, G = NEWROW ? 6 : 2, NA = G * H;
  // (fp32, MC = 32 with the new-row sums: ONE row per lane -- two rows of 64 operands next to 96 fp64 sums
  //  do not fit the register file: 340 bytes of scratch; 4-byte loads are the lesser evil)
  constexpr int V = (NEWROW && MC > 20 && sizeof(T) == 4) ? 1 : RowsPer<T, MC>::V;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const int lane = threadIdx.x & 63;
  const bool hi = lane & 1;  // this lane sums columns [H, MC), its neighbour [0, H)
  const int64_t dy = pe.on ? (int64_t)(((intptr_t)pr - (intptr_t)wy) / (intptr_t)sizeof(T)) : 0;
  const int64_t ds = pe.on ? (int64_t)(((intptr_t)pd - (intptr_t)ws) / (intptr_t)sizeof(T)) : 0;
  auto process = [&](int64_t i, auto wt, auto paired_t) {
    constexpr int W = decltype(wt)::value;
    constexpr bool paired = decltype(paired_t)::value;
    double xv[W], gv[W], rv[W], yf[W], sa[W];
    T a[MC][W], b[MC][W];
    int iw[W];
    ldx<W, NT>(g + i, gv);
    ldx<W, NT>(x + i, xv);
    ldi<W>(iwhere + i, iw);
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
      {
        const double zk = xcp_free<T>(xv[k], gv[k], iw[k], tsum);
        double rr = -theta * (zk - xv[k]) - gv[k];
#pragma unroll
        for (int j = 0; j < MC; ++j)
          if (j < col) rr = rr + (double)a[j][k] * cf.a[j] + (double)b[j][k] * cf.a[MAXM + j];
        rv[k] = iw[k] <= 0 ? rr : 0.0;
      }
      yf[k] = 0.0, sa[k] = 0.0;
      if constexpr (NEWROW) {
        double yn = 0.0, sn = 0.0;
#pragma unroll
        for (int j = 0; j < MC; ++j)
          if (j == col - 1) {
            yn = (double)a[j][k];
            sn = (double)b[j][k];
          }
        yf[k] = iw[k] <= 0 ? yn : 0.0;
        sa[k] = iw[k] <= 0 ? 0.0 : sn;
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      double prv = 0.0, pyf = 0.0, psa = 0.0;
      if constexpr (paired) {
        prv = pair_xchg(rv[k]);
        if constexpr (NEWROW) pyf = pair_xchg(yf[k]), psa = pair_xchg(sa[k]);
      }
#pragma unroll
      for (int jj = 0; jj < H; ++jj) {
        // own rows, own half of the columns
        const T alo = in_reg(a[jj][k]), ahi = in_reg(a[H + jj][k]);
        const T blo = in_reg(b[jj][k]), bhi = in_reg(b[H + jj][k]);
        const double aj = widen_late(hi ? ahi : alo);
        const double bj = widen_late(hi ? bhi : blo);
        acc[jj] += aj * rv[k];
        acc[H + jj] += bj * rv[k];
        if constexpr (NEWROW) {
          acc[2 * H + jj] += yf[k] * aj;
          acc[3 * H + jj] += sa[k] * bj;
          acc[4 * H + jj] += sa[k] * aj;
          acc[5 * H + jj] += bj * yf[k];
        }
        if constexpr (paired) {  // the neighbour's rows: it sends the half it does not sum itself
          const T sa_ = hi ? alo : ahi;
          const T sb_ = hi ? blo : bhi;
          const double paj = widen_late(pair_xchg(sa_));
          const double pbj = widen_late(pair_xchg(sb_));
          acc[jj] += paj * prv;
          acc[H + jj] += pbj * prv;
          if constexpr (NEWROW) {
            acc[2 * H + jj] += pyf * paj;
            acc[3 * H + jj] += psa * pbj;
            acc[4 * H + jj] += psa * paj;
            acc[5 * H + jj] += pbj * pyf;
          }
        }
      }
    }
  };
  // (pairs are complete: lanes 2k and 2k+1 of a wave run the same number of trips, nve is even)
  const int64_t nv = n / V, nve = nv & ~(int64_t)1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; iv < nve; iv += stride)
    process(iv * V, WTag<V>{}, std::true_type{});
  if (blockIdx.x == 0 && threadIdx.x < 2) {  // rows without a neighbour: both lanes, one half each
    if (nv > nve) process(nve * V, WTag<V>{}, std::false_type{});
    for (int64_t rrow = nv * V; rrow < n; ++rrow) process(rrow, WTag<1>{}, std::false_type{});
  }
  // each lane holds the sums of its half of the columns: zeros for the other half, then the ordinary
  // fixed-order reduction over all lanes (slots in the plain kernel's layout: group * MC + column)
  double full[G * MC];
#pragma unroll
  for (int gq = 0; gq < G; ++gq)
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
      full[gq * MC + jj] = hi ? 0.0 : acc[gq * H + jj];
      full[gq * MC + H + jj] = hi ? acc[gq * H + jj] : 0.0;
    }
  block_reduce_store<G * MC>(full, G * MC, 0, 0, part, MAX_BLOCKS);
}

template <typename T>
void launch_cmprlb_wtv(Queue &q, int64_t n, const T *x, const T *g, double tsum,
                       const iw_t *iwhere, WStore<T> w, int head, int col, double theta,
                       const Coef &a, int newrow, const T *pr, const T *pd, Pend pe) {
  const int gr = grid_for_w(q, n, VecOf<T>::V);
  const int mc = maxc_for(col);
  const bool spec = pe.on && col == mc;  // the steady-state shape (pair pending, memory full)
#define LB_PAIRK(MCV, NEWROWV)                                                                          \
  do {                                                                                                  \
    if (q.nt)                                                                                           \
      hipLaunchKernelGGL((cmprlb_wtv_pair_kernel<T, MCV, true, NEWROWV>), dim3(gr), dim3(BLOCK), 0,     \
                         q.stream, n, x, g, tsum, iwhere, w.ws, w.wy, w.ld, w.m, head, col, theta, a,   \
                         pr, pd, pe, q.part());                                                         \
    else                                                                                                \
      hipLaunchKernelGGL((cmprlb_wtv_pair_kernel<T, MCV, false, NEWROWV>), dim3(gr), dim3(BLOCK), 0,    \
                         q.stream, n, x, g, tsum, iwhere, w.ws, w.wy, w.ld, w.m, head, col, theta, a,   \
                         pr, pd, pe, q.part());                                                         \
  } while (0)
  // the plain kernel: MC = 5, 10 always; MC = 20 without the new-row sums, and with them for fp32 in the
  // steady-state shape (the only MC = 20 new-row shape it holds without scratch)
#define LB_CMPRLB(MCV, NEWROWV, PSPECV)                                                              \
  do {                                                                                              \
    constexpr int MC = MCV;                                                                          \
    if (q.nt) {                                                                                     \
      constexpr bool NTV = true;                                                                    \
      DISPATCH_PIPE(MC, hipLaunchKernelGGL((cmprlb_wtv_kernel<T, MC, NEWROWV, NTV, PSPECV, PIPEV>),  \
                                           dim3(gr), dim3(BLOCK), 0, q.stream, n, x, g, tsum,        \
                                           iwhere, w.ws, w.wy, w.zero, w.ld, w.m, head, col, theta,  \
                                           a, pr, pd, pe, q.part()));                                \
    } else {                                                                                        \
      constexpr bool NTV = false;                                                                   \
      DISPATCH_PIPE(MC, hipLaunchKernelGGL((cmprlb_wtv_kernel<T, MC, NEWROWV, NTV, PSPECV, PIPEV>),  \
                                           dim3(gr), dim3(BLOCK), 0, q.stream, n, x, g, tsum,        \
                                           iwhere, w.ws, w.wy, w.zero, w.ld, w.m, head, col, theta,  \
                                           a, pr, pd, pe, q.part()));                                \
    }                                                                                               \
  } while (0)
#define LB_CMPRLB_SMALL(MCV)                       \
  do {                                             \
    if (newrow) {                                  \
      if (spec)                                    \
        LB_CMPRLB(MCV, true, true);                \
      else                                         \
        LB_CMPRLB(MCV, true, false);               \
    } else {                                       \
      if (spec)                                    \
        LB_CMPRLB(MCV, false, true);               \
      else                                         \
        LB_CMPRLB(MCV, false, false);              \
    }                                              \
  } while (0)
  if (mc == 5) {
    LB_CMPRLB_SMALL(5);
  } else if (mc == 10) {
    LB_CMPRLB_SMALL(10);
  } else if (mc == 20) {
    if (newrow) {
      if constexpr (sizeof(T) == 4) {
        if (spec)
          LB_CMPRLB(20, true, true);
        else
          LB_PAIRK(20, true);
      } else {
        LB_PAIRK(20, true);
      }
    } else if (spec) {
      LB_CMPRLB(20, false, true);
    } else {
      LB_CMPRLB(20, false, false);
    }
  } else {
    // col 21..32: always the pair-shared kernel (with the new-row sums: formk stays incremental there too)
    if (newrow)
      LB_PAIRK(32, true);
    else
      LB_PAIRK(32, false);
  }
#undef LB_CMPRLB_SMALL
#undef LB_CMPRLB
#undef LB_PAIRK
  LB_LAUNCHED(q);
  launch_finalize(q, gr, (newrow ? 6 : 2) * mc, 0, 0);
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_cmprlb_wtv<T>(Queue &, int64_t, const T *, const T *, double, const iw_t *, WStore<T>, int, int, double, const Coef &, int, const T *, const T *, Pend);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
