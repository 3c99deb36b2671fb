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
  const int gr = grid_for_w(q, n, VecOf<T>::V);
  const int nold = col - 1;
  DISPATCH_MAXC_NT(nold, q.nt, hipLaunchKernelGGL((update_pairs_kernel<T, MC, NTV>), dim3(gr), dim3(BLOCK), 0,
                                         q.stream, n, g, r, d, stp, w.ws, w.wy, w.zero, w.ld, w.m,
                                         head, nold, itail, q.part()));
  LB_LAUNCHED(q);
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
// One term of a running dot product.  fp64 kinds: product and sum rounded separately, as the reference's ddot
// rounds them (src/lbfgsb_blas_module.F90:187-202) -- only the ORDER of the sum differs from the reference.  REAL32:
// the operands are fp32 values (after a unit step; s = stp d is a double otherwise), their product is exact in
// fp64, so the fused form is the SAME number with one instruction less -- and the fp32 m = 20 instantiation of
// this pass is issue-bound, not HBM-bound.
// FUSED (the pair-shared pass on the tile-local layout of W): that pass is bound by what it issues next to its loads,
// and its sums are added up in an order of their own anyway (compared at 1e-10, like every reduction): one rounding
// per term instead of two.
template <typename T, bool FUSED = false>
__device__ __forceinline__ double dot_term(double acc, double a, double b) {
  if constexpr (sizeof(T) == 4 || FUSED)
    return __builtin_fma(a, b, acc);
  else
    return acc + a * b;
}
template <typename T>
struct UpdScanCtx {
  const T *x, *l, *u, *g, *r, *d, *ws, *wy, *zero;
  const nb_t *nbd;
  const iw_t *iwhere;
  int64_t ldw;
  int m, head, nold;
  // uniform bounds (bit 0: every l_i is the same value, bit 1: every u_i, bit 2: every nbd_i): the
  // array pointer then is a 64-byte buffer holding that value and every lane reads ITS start -- a
  // cache hit instead of an HBM stream, selected by ADDRESS like the zero columns (no new code path)
  int ub;
  // PAIR: column pointers per lane parity, in LDS (see UpdScanPairTrip)
  const unsigned long long *ptab;
};
template <typename T, int MC, int W>
struct UpdScanRegs {  // the register images of one row group
  RawOf<T, W> rx, rl, ru, rg, rr, rd, ra[MC], rb[MC];
  RawOf<nb_t, W> rnb;
  RawOf<iw_t, W> riw;
};
template <typename T, int MC, int W, bool NT>
struct UpdScanTrip : UpdScanRegs<T, MC, W>, NaturalRows<W> {
  static constexpr int NL = 8 + 2 * MC;
  __device__ __forceinline__ void issue(const UpdScanCtx<T> &c, int64_t i) {
    constexpr int B = (int)sizeof(T) * W;
    raw_issue<B, NT>(this->rx, c.x + i);
    raw_issue<B, NT>(this->rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(this->ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(this->rg, c.g + i);
    raw_issue<B, NT>(this->rr, c.r + i);
    raw_issue<B, NT>(this->rd, c.d + i);
    raw_issue<W, false>(this->rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<W, false>(this->riw, c.iwhere + i);
    issue_cols<T, MC, W, NT>(c.wy, c.ws, (const T *)nullptr, (const T *)nullptr, c.zero, i, c.nold, c.head,
                             c.m, c.ldw, Pend{0, 1.0}, this->ra, this->rb);
  }
  __device__ __forceinline__ void land() {}
};
// The tile-local free-row layout of W (for_tiles_cw, device_util.hpp).  The n-vector operands of a row come from the
// row itself, its W entries from its slot -- if its layout bit is set; the rows behind the layout-free ones are
// multiplied by exact zeros in nearly every case (s = 0: they sat at a bound, -g masked: they stay there) and read
// the zero buffer instead; the few that changed status since the layout was made fetch their entries in the kernel
// body (reload_cols), so the sums never depend on the layout.
// ... one row of the partial tile:
template <typename T, int MC, bool NT>
struct UpdScanTripCW1 : UpdScanRegs<T, MC, 1>, CwOneRow {
  static constexpr int NL = 8 + 2 * MC;
  // `first`: the first slot of the row's tile.  A row whose layout bit is clear reads THAT entry instead of its own
  // (a line the tile's front run brings in anyway): no predicate, no second address.  Its value is never used where
  // it could matter -- such a row either contributes products with exact zeros (it did not move, it is not free: any
  // finite factor will do) or is caught as `miss` in the kernel body and fetches its own entries (reload_cols).
  __device__ __forceinline__ void issue_cw(const UpdScanCtx<T> &c, int64_t i, int64_t slot, bool lf, int64_t first) {
    constexpr int B = (int)sizeof(T);
    this->ri_ = i, this->ws_ = slot, this->lf_ = lf;
    raw_issue<B, NT>(this->rx, c.x + i);
    raw_issue<B, NT>(this->rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(this->ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(this->rg, c.g + i);
    raw_issue<B, NT>(this->rr, c.r + i);
    raw_issue<B, NT>(this->rd, c.d + i);
    raw_issue<1, false>(this->rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<1, false>(this->riw, c.iwhere + i);
    cols<false>(c, lf ? slot : first);  // (plain loads: see UpdScanPairTripCW)
  }
  template <bool NTL>
  __device__ __forceinline__ void cols(const UpdScanCtx<T> &c, int64_t srow) {
    constexpr int B = (int)sizeof(T);
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, c.nold, c.head, c.m, c.ldw) + srow;
      const bool live = j < c.nold;
      raw_issue<B, NTL>(this->ra[j], live ? c.wy + off : c.zero);
      raw_issue<B, NTL>(this->rb[j], live ? c.ws + off : c.zero);
    }
  }
  __device__ __forceinline__ void land() {}
  // Into registers of their own, merged afterwards: a load INTO ra / rb inside this (rare) branch would make every
  // later use of them wait for whatever is in flight -- with two trips in flight, the next trip's loads.
  __device__ __forceinline__ void reload_cols(const UpdScanCtx<T> &c, const bool (&miss)[1]) {
    constexpr int B = (int)sizeof(T);
    RawOf<T, 1> ty[MC], ts[MC];
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, c.nold, c.head, c.m, c.ldw) + (miss[0] ? this->ws_ : this->ri_ & ~(int64_t)127);
      const bool live = j < c.nold;
      raw_issue<B, false>(ty[j], live ? c.wy + off : c.zero);
      raw_issue<B, false>(ts[j], live ? c.ws + off : c.zero);
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      this->ra[j].v = miss[0] ? ty[j].v : this->ra[j].v;
      this->rb[j].v = miss[0] ? ts[j].v : this->rb[j].v;
    }
  }
};
// ... the rows lane, lane + 64 of a full tile as one row group of width 2 (fp64: two 8-byte halves per register
// image).  Addresses are a wave-uniform base (array + first row of the tile) + a 32-bit lane offset.  Unroll slots
// beyond the stored pairs read the zero buffer (>= a tile long).
template <typename T, int MC, bool NT>
struct UpdScanTripCW2 : UpdScanRegs<T, MC, 2>, CwPairRows {
  static_assert(sizeof(T) == 8, "compact W: fp64");
  static constexpr int NL = 2 * (8 + 2 * MC);
  RawReg<1> nb_[2], iw_[2];
  // (addresses as the natural-order kernels form them: ONE 64-bit row offset per lane and row, the uniform column
  //  offset added per load -- 40 uniform column bases would not fit the scalar registers)
  template <bool NTL>
  __device__ __forceinline__ void cols_row(const UpdScanCtx<T> &c, int k, int64_t srow) {
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, c.nold, c.head, c.m, c.ldw) + srow;
      const bool live = j < c.nold;
      raw_issue_half<NTL>(this->ra[j], k, live ? c.wy + off : c.zero);
      raw_issue_half<NTL>(this->rb[j], k, live ? c.ws + off : c.zero);
    }
  }
  __device__ __forceinline__ void issue_cw(const UpdScanCtx<T> &c, const CwTile &t) {
    this->tile_ = t;
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t i = t.tb + (lane + 64 * k);
      raw_issue_half<NT>(this->rx, k, c.x + i);
      raw_issue_half<NT>(this->rl, k, (c.ub & 1) ? c.l : c.l + i);
      raw_issue_half<NT>(this->ru, k, (c.ub & 2) ? c.u : c.u + i);
      raw_issue_half<NT>(this->rg, k, c.g + i);
      raw_issue_half<NT>(this->rr, k, c.r + i);
      raw_issue_half<NT>(this->rd, k, c.d + i);
      raw_issue<1, false>(nb_[k], (c.ub & 4) ? c.nbd : c.nbd + i);
      raw_issue<1, false>(iw_[k], c.iwhere + i);
    }
    int sl[2];
    bool lf[2];
    cw_slots(t, sl, lf);
    // A row whose layout bit is clear reads the tile's FIRST entry instead of its own (a line the tile's front run
    // brings in anyway): no predicate, no second address.  Its value is never used where it could matter -- such a
    // row either contributes products with exact zeros (it did not move, it is not free: any finite factor will do)
    // or is caught as `miss` in the kernel body and fetches its own entries (reload_cols).
    cols_row<false>(c, 0, t.tb + (lf[0] ? sl[0] : 0));  // (plain loads: see SubsmTripCW2)
    cols_row<false>(c, 1, t.tb + (lf[1] ? sl[1] : 0));
  }
  __device__ __forceinline__ void land() {
    raw_join_bytes(this->rnb, nb_[0], nb_[1]);
    raw_join_bytes(this->riw, iw_[0], iw_[1]);
  }
  __device__ __forceinline__ void reload_cols(const UpdScanCtx<T> &c, const bool (&miss)[2]) {
    int sl[2];
    bool lf[2];
    cw_slots(this->tile_, sl, lf);
    // (into registers of their own, merged afterwards: see UpdScanTripCW1)
    RawReg<8> ty[MC][2], ts[MC][2];
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, c.nold, c.head, c.m, c.ldw) + this->tile_.tb;
      const bool live = j < c.nold;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        raw_issue<8, false>(ty[j][k], live ? c.wy + off + (miss[k] ? sl[k] : 0) : c.zero);
        raw_issue<8, false>(ts[j][k], live ? c.ws + off + (miss[k] ? sl[k] : 0) : c.zero);
      }
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      if (miss[0]) this->ra[j].v.xy = ty[j][0].v, this->rb[j].v.xy = ts[j][0].v;
      if (miss[1]) this->ra[j].v.zw = ty[j][1].v, this->rb[j].v.zw = ts[j][1].v;
    }
  }
};
// PAIR (see update_scan_kernel): the lanes 2p, 2p + 1 of a wave work on the 2 W rows [i0, i0 + 2 W) together.
// Each lane owns W of the rows for everything a row needs once (x, g, bounds, the n-loop of cauchy), and HALF of
// the columns over all 2 W rows: the even lane logical columns [0, MC/2), the odd lane [MC/2, MC) -- so a lane
// loads 16 bytes of each of ITS columns (the rows of both lanes) and no column value ever crosses lanes; a wave's
// load instruction reads two columns, 512 contiguous bytes of each.  Only the row scalars (s, -g, the masked
// y and s of formk's new row) are exchanged, by DPP.
// The per-lane column pointers (2 x MC of them; unroll slots beyond the stored pairs point at the zero buffer)
// live in LDS, read 16 bytes per column pair and trip: as 2 MC uniform 64-bit offsets they do not fit the scalar
// registers next to everything else -- the compiler kept them in VGPR lanes and fetched them back with ~90
// v_readlane per trip (profiles/round5_u_pair_isa.md).
template <typename T, int MC, int W, bool NT>
struct UpdScanPairTrip : NaturalRows<W> {
  static constexpr int H = MC / 2;
  static constexpr int NL = 8 + 2 * H;
  RawOf<T, W> rx, rl, ru, rg, rr, rd;
  RawOf<T, 2 * W> ra[H], rb[H];
  RawOf<nb_t, W> rnb;
  RawOf<iw_t, W> riw;
  __device__ __forceinline__ void issue(const UpdScanCtx<T> &c, int64_t i) {
    constexpr int B = (int)sizeof(T) * W;
    const bool hi = threadIdx.x & 1;
    const int64_t i0 = i - (hi ? W : 0);  // first row of the lane pair (a multiple of 2 W: 16-byte aligned)
    // (the table is the same in every trip: an opaque index keeps its 2 MC pointers from being hoisted out of
    //  the row loop into registers that are not there.  Read FIRST: the row vectors' address arithmetic and
    //  loads then cover the LDS latency, which a wave that runs alone on its SIMD would otherwise sit out)
    int tb = hi ? 2 * H : 0;
    asm volatile("" : "+v"(tb));
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 p[H];
#pragma unroll
    for (int jj = 0; jj < H; ++jj) p[jj] = *reinterpret_cast<const u64x2 *>(c.ptab + tb + 2 * jj);
    __builtin_amdgcn_sched_barrier(0);
    raw_issue<B, NT>(rx, c.x + i);
    raw_issue<B, NT>(rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(rg, c.g + i);
    raw_issue<B, NT>(rr, c.r + i);
    raw_issue<B, NT>(rd, c.d + i);
    raw_issue<W, false>(rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<W, false>(riw, c.iwhere + i);
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
      const int64_t o = (hi && H + jj >= c.nold) ? 0 : i0;  // the zero buffer is 256 bytes, not a column
      // (pointers that come out of LDS: say that they point to global memory, or the loads are flat ones)
      typedef const __attribute__((address_space(1))) T *gptr;
      raw_issue<2 * B, NT>(ra[jj], (const T *)((gptr)p[jj].x + o));
      raw_issue<2 * B, NT>(rb[jj], (const T *)((gptr)p[jj].y + o));
    }
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
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
      raw_land(ra[jj]);
      raw_land(rb[jj]);
    }
  }
};
// The same sharing under the tile-local free-row layout of W (one row per lane and trip: for_halves_cw).  A pair's
// two rows are not neighbours in a column any more: each lane reads its columns at its OWN row's slot and at its
// partner's (two 8-byte loads into the halves of the register image).  The pair is (l, l + 32), not (2p, 2p + 1):
// a load instruction then reads the slots of 32 CONSECUTIVE rows of two columns -- two contiguous pieces -- instead
// of every other slot of 64 rows, whose lines the partner instruction has to touch again (bare kernels: - 10 % at
// nfree/n = 0.5, profiles/round6_g_compact_shapes_pair32.txt); slots and row scalars cross by v_permlane32_swap.
// What the sharing buys here is the register file: 8 x MC/2 column sums per lane instead of 8 x MC leave the 95 fp64
// accumulators of the MC = 10 new-row pass in VGPRs (no accumulator lives in the AGPR file, which cost two moves per
// update), and the pass is bound by its loads again instead of by what it issues.
// TIGHT: the caller guarantees nold == MC - 1 (every iteration once the memory is full): the one dead unroll slot
// is the odd lane's last, known at compile time -- no select in front of the other loads.
template <typename T, int MC, bool NT, bool TIGHT>
struct UpdScanPairTripCW : CwOneRow {
  static_assert(sizeof(T) == 8, "compact W: fp64");
  static constexpr int H = MC / 2;
  static constexpr int NL = 8 + 4 * H;
  RawOf<T, 1> rx, rl, ru, rg, rr, rd;
  RawOf<T, 2> ra[H], rb[H];
  RawOf<nb_t, 1> rnb;
  RawOf<iw_t, 1> riw;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) T *gptr;
  // this lane's columns (H pairs from the LDS table) for the pair's rows at a0 (even lane's row), a1 (odd lane's)
  template <bool NTL, typename R>
  __device__ __forceinline__ void cols(const UpdScanCtx<T> &c, int64_t a0, int64_t a1, R (&qa)[H], R (&qb)[H]) {
    const bool hi = threadIdx.x & 32;
    int tb = hi ? 2 * H : 0;
    asm volatile("" : "+v"(tb));
    u64x2 p[H];
#pragma unroll
    for (int jj = 0; jj < H; ++jj) p[jj] = *reinterpret_cast<const u64x2 *>(c.ptab + tb + 2 * jj);
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
      const bool dead = TIGHT ? (jj == H - 1 && hi) : (hi && H + jj >= c.nold);  // (that table entry is the zero buffer)
      raw_issue_half<NTL>(qa[jj], 0, (const T *)((gptr)p[jj].x + (dead ? 0 : a0)));
      raw_issue_half<NTL>(qa[jj], 1, (const T *)((gptr)p[jj].x + (dead ? 0 : a1)));
      raw_issue_half<NTL>(qb[jj], 0, (const T *)((gptr)p[jj].y + (dead ? 0 : a0)));
      raw_issue_half<NTL>(qb[jj], 1, (const T *)((gptr)p[jj].y + (dead ? 0 : a1)));
    }
  }
  __device__ __forceinline__ void issue_cw(const UpdScanCtx<T> &c, int64_t i, int64_t slot, bool lf, int64_t first) {
    constexpr int B = (int)sizeof(T);
    this->ri_ = i, this->ws_ = slot, this->lf_ = lf;
    raw_issue<B, NT>(rx, c.x + i);
    raw_issue<B, NT>(rl, (c.ub & 1) ? c.l : c.l + i);
    raw_issue<B, NT>(ru, (c.ub & 2) ? c.u : c.u + i);
    raw_issue<B, NT>(rg, c.g + i);
    raw_issue<B, NT>(rr, c.r + i);
    raw_issue<B, NT>(rd, c.d + i);
    raw_issue<1, false>(rnb, (c.ub & 4) ? c.nbd : c.nbd + i);
    raw_issue<1, false>(riw, c.iwhere + i);
    // (a row whose layout bit is clear reads the first entry of its tile: see UpdScanTripCW1)
    int64_t a0, a1;  // the slots of the pair's rows: the lane below 32, the lane above
    half_pair(lf ? slot : first, a0, a1);
    // (plain loads for the W entries, whatever NT says for the row vectors: the line in which this half's run of a
    //  column ends is the one the other half's run begins in -- read by another wave at about the same time, and
    //  a nontemporal line is fetched from HBM for each of them: 2.58 -> 2.39 ms at n = 1e8)
    cols<false>(c, a0, a1, ra, rb);
  }
  __device__ __forceinline__ void land() {}
  // miss[0]: THIS lane's row needs its own entries.  Both lanes of a pair fetch their columns of that row (into
  // registers of their own, merged afterwards: see UpdScanTripCW1).
  __device__ __forceinline__ void reload_cols(const UpdScanCtx<T> &c, const bool (&miss)[1]) {
    int mi0, mi1;
    half_pair((int)miss[0], mi0, mi1);
    const bool m0 = mi0 != 0, m1 = mi1 != 0;
    const int64_t first = this->ri_ & ~(int64_t)127;
    int64_t a0, a1;
    half_pair((miss[0] || this->lf_) ? this->ws_ : first, a0, a1);
    RawOf<T, 2> ta[H], tb2[H];
    cols<false>(c, a0, a1, ta, tb2);
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
      if (m0) ra[jj].v.xy = ta[jj].v.xy, rb[jj].v.xy = tb2[jj].v.xy;
      if (m1) ra[jj].v.zw = ta[jj].v.zw, rb[jj].v.zw = tb2[jj].v.zw;
    }
  }
};
// NEWROW: the pass also yields the new row/column of formk's WN1 (:1756-1793) for the pair being
// formed here -- with the free/active split of the rows as cauchy's n-loop leaves it, i.e.
// BEFORE the breakpoint walk (the host corrects the sums for the few rows the walk fixes, from
// the records it gathers for them anyway).  With them and the walk's own p = W'd the host has
// W'Z r in closed form (solver.hip, subspace_closed_form), and the iteration needs no third pass
// over W.  Extra sum slots, X = 4 MC + 9:  [X, X+MC) sum_free y Wy_j | [X+MC, ..) sum_act s Ws_j |
// [X+2MC, ..) sum_act s Wy_j | [X+3MC, ..) sum_free Ws_j y | [X+4MC ..+4) sum_free y y,
// sum_act s s, sum_act s y, sum_free s y   (y, s in their stored form); min and max follow.
// PAIR (MC = 20 with the new-row sums): the 8 sums per column are what fills the register
// file (175 fp64 accumulators per lane), and a kernel that large runs one wave per SIMD with a
// single trip in flight.  Neighbouring lanes therefore SHARE the per-column accumulators: the even
// lane sums columns [0, 10), the odd lane columns [10, 20), each over the rows of BOTH lanes, in
// row order -- each lane loads its own columns for both lanes' rows (UpdScanPairTrip) and the row
// scalars of the other lane come over by DPP -- which leaves room for the second trip in flight.
// The caller passes a row count that is a multiple of 2 V (pairs are always complete) and runs the
// plain instantiation on the few rows that remain.
// CW: W in the tile-local free-row layout `lmask` (fp64, MC <= 10; for_tiles_cw)
template <typename T, int MC, bool NT, bool PIPE, bool NEWROW, bool PAIR = false, bool CW = false, bool TIGHT = false>
__global__ __launch_bounds__(BLOCK) void update_scan_kernel(
    int64_t n, const T *__restrict__ x, const T *__restrict__ l, const T *__restrict__ u,
    const nb_t *__restrict__ nbd, const T *__restrict__ g, const T *__restrict__ r,
    const T *__restrict__ d, int dimpl, double stp, iw_t *iwhere, T *tbrk, T *ws, T *wy,
    const T *__restrict__ zero, int64_t ldw, int m, int head, int nold, int itail, int store_pair,
    int store_iw, double cand_hi, uint64_t *ckeys, uint32_t *cidx, uint32_t ccap, uint32_t *ccount,
    int ub, double *part, const uint64_t *__restrict__ lmask = nullptr) {
  static_assert(!CW || (sizeof(T) == 8 && MC <= 10 && !PIPE), "compact W: fp64, MC <= 10, one trip in flight");
  constexpr int NX = NEWROW ? 4 * MC + 4 : 0;  // extra sums
  constexpr int X = 4 * MC + 9;                // first extra slot
  constexpr int NA = 4 * MC + 11 + NX;
  constexpr int IMIN = X + NX, IMAX = X + NX + 1;  // bkmin, |proj g|
  // (fp32, MC = 10 with the new-row sums: 4 rows per lane after all -- the 16-byte loads are worth
  //  more than the second wave the 512 registers cost: 1.88 -> 1.65 ms at n = 1e8)
  constexpr int V = CW ? 1 : ((sizeof(T) == 4 && MC == 10 && NEWROW) ? 4 : RowsPerAcc<T, MC, NA>::V);
  static_assert(!PAIR || (NEWROW && MC % 2 == 0), "PAIR: the new-row instantiation, even MC");
  constexpr int H = MC / 2;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  acc[IMIN] = LB_INF;
  double accp[PAIR ? 8 : 1][PAIR ? H : 1];  // PAIR: this lane's half of the columns, 8 sums each
#pragma unroll
  for (int a = 0; a < (PAIR ? 8 : 1); ++a)
#pragma unroll
    for (int b = 0; b < (PAIR ? H : 1); ++b) accp[a][b] = 0.0;
  const bool hi = threadIdx.x & ((PAIR && CW) ? 32 : 1);  // (which half of the columns: see the two pair trips)
  const int64_t offn = (int64_t)(itail - 1) * ldw;
  __shared__ unsigned long long ptab[PAIR ? 2 * MC : 1];  // [lane parity][column of the half][Wy, Ws]
  if constexpr (PAIR) {
    if (threadIdx.x < 2 * MC) {
      const int j = (int)threadIdx.x >> 1;  // logical column: the odd lanes' half follows the even lanes'
      const T *base = (threadIdx.x & 1) ? ws : wy;
      ptab[threadIdx.x] = (unsigned long long)(uintptr_t)(j < nold ? base + col_off(j, nold, head, m, ldw) : zero);
    }
    __syncthreads();
  }
  const UpdScanCtx<T> ctx{x, l, u, g, r, d, ws, wy, zero, nbd, iwhere, ldw, m, head, nold, ub, ptab};
  __shared__ T dict[16];
  dict_fill<T>(dict, l, u, ub);
  using TripV = std::conditional_t<PAIR, UpdScanPairTrip<T, MC, V, NT>, UpdScanTrip<T, MC, V, NT>>;
  using Trip1 = std::conditional_t<PAIR, UpdScanPairTrip<T, MC, 1, NT>, UpdScanTrip<T, MC, 1, NT>>;
  auto body = [&](auto &tr, int64_t i, auto wt) {
    constexpr int W = decltype(wt)::value;
    double xv[W], lv[W], uv[W], gv[W], rv[W], dv[W], tb[W], ng[W];
    int nb[W], iw[W];
    raw_get<W>(tr.rx, (const T *)nullptr, xv);
    raw_get<W>(tr.rl, (const T *)nullptr, lv);
    raw_get<W>(tr.ru, (const T *)nullptr, uv);
    raw_get<W>(tr.rg, (const T *)nullptr, gv);
    raw_get<W>(tr.rr, (const T *)nullptr, rv);
    raw_get<W>(tr.rd, (const T *)nullptr, dv);
    raw_geti<W>(tr.rnb, (const nb_t *)nullptr, nb);
    raw_geti<W>(tr.riw, (const iw_t *)nullptr, iw);
    dict_apply<T, W>(dict, ub, nb, lv, uv);
    bool iw_changed = false;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      // (dimpl: `d` points at t, the previous iterate, and the direction is x - t: the unit
      //  first trial step of a lean subsm_update_kernel pass, see Pend::impl)
      if (dimpl) dv[k] = (double)(T)(xv[k] - dv[k]);
      // ---- the line search's own sums at this trial point: g'd (:2244), |proj g| (:781) ----
      acc[4 * MC + 7] = acc[4 * MC + 7] + gv[k] * dv[k];
      acc[IMAX] = fmax(acc[IMAX], proj_g(xv[k], lv[k], uv[k], nb[k], gv[k]));
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
          acc[IMIN] = fmin(acc[IMIN], tb[k]);
        } else if (nb[k] >= 2 && neggi > 0.0) {
          tb[k] = tu / neggi;
          acc[4 * MC + 4] += 1.0;
          acc[IMIN] = fmin(acc[IMIN], tb[k]);
        } else {
          tb[k] = LB_INF;
          acc[4 * MC + 5] += 1.0;
          if (fabs(neggi) > 0.0) acc[4 * MC + 6] += 1.0;
        }
      }
      acc[3 * MC + 1] += rv[k] * ng[k];  // new Wy column . d
      acc[4 * MC + 2] += dv[k] * ng[k];  // new Ws column . d
    }
    // (columns are widened one pair at a time, where they are used: the fp32 instantiations with
    //  many accumulators cannot keep all 2 MC operands of a trip as doubles)
    double yf[W], sa[W];
    if constexpr (NEWROW) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const double yst = (double)(T)rv[k], sst = (double)(T)dv[k];  // as stored in W
        const bool fr = iw[k] <= 0;                                   // after the n-loop's update
        yf[k] = fr ? yst : 0.0;
        sa[k] = fr ? 0.0 : sst;
        acc[X + 4 * MC + 0] = __builtin_fma(yf[k], yst, acc[X + 4 * MC + 0]);
        acc[X + 4 * MC + 1] = __builtin_fma(sa[k], sst, acc[X + 4 * MC + 1]);
        acc[X + 4 * MC + 2] = __builtin_fma(sa[k], yst, acc[X + 4 * MC + 2]);
        acc[X + 4 * MC + 3] = __builtin_fma(fr ? sst : 0.0, yst, acc[X + 4 * MC + 3]);
      }
    }
    if constexpr (std::remove_reference_t<decltype(tr)>::CW) {
      // a row whose W entries are needed (it moved, or it is free at this point) although its layout bit is clear
      bool miss[W], any = false;
#pragma unroll
      for (int k = 0; k < W; ++k) miss[k] = !tr.lf(k) && (dv[k] != 0.0 || iw[k] <= 0), any = any || miss[k];
      if (__ballot(any) != 0ull) tr.reload_cols(ctx, miss);
    }
    auto row_stores = [&]() {
      if (store_pair) {  // else the pair stays pending (see Pend)
        tr.template st_w<false>(ws + offn, i, dv);
        tr.template st_w<false>(wy + offn, i, rv);
      }
      // iwhere settles after the first iterations: store only from waves that changed a row
      if (store_iw && __ballot(iw_changed) != 0ull) tr.sti_rows(iwhere, i, iw);
      if (tbrk) tr.template st_rows<false>(tbrk, i, tb);  // nullptr: the walk recomputes the times it needs
    };
    // PAIR: in front of the column sums, so that nothing of the row part stays live across them (404 bytes
    // of scratch otherwise); the other instantiations hand breakpoints over below and keep them at the end
    if constexpr (PAIR) row_stores();
    if constexpr (PAIR) {
      // the row scalars of the pair's 2 W rows in row order: [0, W) are the even lane's, [W, 2 W) the odd lane's
      // (-g, y and s as stored are values of T and travel as such: one register each for fp32)
      double qd[2 * W], qn[2 * W], qy[2 * W], qs[2 * W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const T n_ = (T)ng[k], y_ = (T)yf[k], s_ = (T)sa[k];
        if constexpr (CW) {  // (W = 1; the pair is (l, l + 32): both values in both lanes, no select)
          T n0, n1, y0, y1, s0, s1;
          half_pair(dv[k], qd[k], qd[W + k]);
          half_pair(n_, n0, n1);
          half_pair(y_, y0, y1);
          half_pair(s_, s0, s1);
          qn[k] = (double)n0, qn[W + k] = (double)n1;
          qy[k] = (double)y0, qy[W + k] = (double)y1;
          qs[k] = (double)s0, qs[W + k] = (double)s1;
        } else {
          const double od = pair_xchg(dv[k]);
          qd[k] = hi ? od : dv[k], qd[W + k] = hi ? dv[k] : od;
          const T on = pair_xchg(n_), oy = pair_xchg(y_), os = pair_xchg(s_);
          qn[k] = (double)(hi ? on : n_), qn[W + k] = (double)(hi ? n_ : on);
          qy[k] = (double)(hi ? oy : y_), qy[W + k] = (double)(hi ? y_ : oy);
          qs[k] = (double)(hi ? os : s_), qs[W + k] = (double)(hi ? s_ : os);
        }
      }
#pragma unroll
      for (int jj = 0; jj < H; ++jj) {
        double aj[2 * W], bj[2 * W];  // this lane's column pair jj (+ H on the odd lane), the pair's rows
        raw_get_col<2 * W, false>(tr.ra[jj], (const T *)nullptr, aj);
        raw_get_col<2 * W, false>(tr.rb[jj], (const T *)nullptr, bj);
#pragma unroll
        for (int k = 0; k < 2 * W; ++k) {
          accp[0][jj] = dot_term<T, CW>(accp[0][jj], qd[k], aj[k]);
          accp[1][jj] = dot_term<T, CW>(accp[1][jj], bj[k], qd[k]);
          accp[2][jj] = dot_term<T, CW>(accp[2][jj], aj[k], qn[k]);
          accp[3][jj] = dot_term<T, CW>(accp[3][jj], bj[k], qn[k]);
          accp[4][jj] = __builtin_fma(qy[k], aj[k], accp[4][jj]);
          accp[5][jj] = __builtin_fma(qs[k], bj[k], accp[5][jj]);
          accp[6][jj] = __builtin_fma(qs[k], aj[k], accp[6][jj]);
          accp[7][jj] = __builtin_fma(bj[k], qy[k], accp[7][jj]);
        }
      }
    } else
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      double aj[W], bj[W];
      raw_get_col<W, false>(tr.ra[j], (const T *)nullptr, aj);
      raw_get_col<W, false>(tr.rb[j], (const T *)nullptr, bj);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += dv[k] * aj[k];               // Sy(col,j) (:2335)
        acc[MC + j] += bj[k] * dv[k];          // Ss(j,col) (:2336)
        acc[2 * MC + 1 + j] += aj[k] * ng[k];  // p_j        (:1301)
        acc[3 * MC + 2 + j] += bj[k] * ng[k];  // p_{col+j}  (:1302)
      }
      if constexpr (NEWROW) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
          acc[X + j] = __builtin_fma(yf[k], aj[k], acc[X + j]);
          acc[X + MC + j] = __builtin_fma(sa[k], bj[k], acc[X + MC + j]);
          acc[X + 2 * MC + j] = __builtin_fma(sa[k], aj[k], acc[X + 2 * MC + j]);
          acc[X + 3 * MC + j] = __builtin_fma(bj[k], yf[k], acc[X + 3 * MC + j]);
        }
      }
    }
    // Breakpoints up to cand_hi -- where the NEXT walk is expected to end, the caller's guess from
    // the previous one -- are handed over with this pass (appended, unordered, like the window
    // kernels do): the usual short walk then needs no window pass and no host sync of its own.
    // (not in the instantiation that has no register to spare for it: NEWROW at MC = 20)
    if (!(NEWROW && MC > 10) && cand_hi >= 0.0) {
      unsigned bits = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const double tr_ = (double)(T)tb[k];  // as brk_time / tbrk hold it
        bits |= (tr_ >= 0.0 && tr_ <= cand_hi) ? (1u << k) : 0u;
      }
      // (once the list is full nothing more is appended: a guess that was far too wide -- the
      //  host then falls back to the window pass -- must not turn this pass into an atomics
      //  benchmark; the count is then only a lower bound, which is all the host needs)
      if (__ballot(bits != 0) != 0ull && __builtin_nontemporal_load(ccount) <= ccap) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const bool pred = (bits >> k) & 1u;
          const unsigned long long mask = __ballot(pred);
          if (mask == 0ull) continue;
          const int leader = __ffsll((long long)mask) - 1;
          uint32_t base = 0;
          if (lane == leader) base = atomicAdd(ccount, (uint32_t)__popcll(mask));
          base = __shfl(base, leader);
          if (pred) {
            const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (pos < ccap) {
              ckeys[pos] = key_of((double)(T)tb[k]);
              cidx[pos] = (uint32_t)tr.row(i, k);
            }
          }
        }
      }
    }
    if constexpr (!PAIR) row_stores();
  };
  if constexpr (CW && PAIR)  // (the caller passes whole tiles)
    for_halves_cw<UpdScanPairTripCW<T, MC, NT, TIGHT>>(n, ctx, lmask, body);
  else if constexpr (CW && NEWROW)  // (one row per lane and trip, two trips in flight: see for_halves_cw)
    for_halves_cw<UpdScanTripCW1<T, MC, NT>>(n, ctx, lmask, body);
  else if constexpr (CW)
    for_tiles_cw<UpdScanTripCW2<T, MC, NT>, UpdScanTripCW1<T, MC, NT>>(n, ctx, lmask, body);
  else
    for_rows_raw<TripV, Trip1, V, PIPE, 0>(n, ctx, body);
  if constexpr (PAIR) {
    // each lane holds the sums of its half of the columns: zeros for the other half, then the
    // ordinary fixed-order reduction over all lanes
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
      const double z = 0.0;
      acc[jj] = hi ? z : accp[0][jj], acc[H + jj] = hi ? accp[0][jj] : z;
      acc[MC + jj] = hi ? z : accp[1][jj], acc[MC + H + jj] = hi ? accp[1][jj] : z;
      acc[2 * MC + 1 + jj] = hi ? z : accp[2][jj], acc[2 * MC + 1 + H + jj] = hi ? accp[2][jj] : z;
      acc[3 * MC + 2 + jj] = hi ? z : accp[3][jj], acc[3 * MC + 2 + H + jj] = hi ? accp[3][jj] : z;
      acc[X + jj] = hi ? z : accp[4][jj], acc[X + H + jj] = hi ? accp[4][jj] : z;
      acc[X + MC + jj] = hi ? z : accp[5][jj], acc[X + MC + H + jj] = hi ? accp[5][jj] : z;
      acc[X + 2 * MC + jj] = hi ? z : accp[6][jj], acc[X + 2 * MC + H + jj] = hi ? accp[6][jj] : z;
      acc[X + 3 * MC + jj] = hi ? z : accp[7][jj], acc[X + 3 * MC + H + jj] = hi ? accp[7][jj] : z;
    }
  }
  block_reduce_store<NA>(acc, X + NX, 1, 1, part, MAX_BLOCKS);
}
// ---- more than 20 old pairs with formk's new-row sums: the split pass ----
// Eight sums per column pair are 256 fp64 accumulators at MC = 32 -- no sharing between two lanes brings that
// into the register file next to the 64 operands of a row.  The pass runs as SEVERAL launches of the MC <= 20
// kernels over <= 16 columns each (a sub-range of the ring of pairs is again a ring: head' = head + j0):
// W is still read once, the row vectors (x, g, r, d, bounds: ~10 % of the bytes at two parts) once per part --
// against a third pass over W (cmprlb_wtv) in the iteration, which this saves.  The sums that do not depend on
// the columns (y'y, the line search's g'd, cauchy's f1 and counts, the new pair's own dots) are taken from the
// first launch, which also stores iwhere; the others find iwhere already final (the n-loop's update is
// idempotent).  update_split_merge_kernel puts the results into the layout of ONE launch with column
// capacity MO = maxc_stride(nold) -- a stride of the result layout only, so any number of pairs is served
// (m > 32: solver.hip runs this pass in front of the unfused subspace steps).
struct SplitPlan {
  int nparts, mo;
  int src[SPLIT_MAXPARTS], j0[SPLIT_MAXPARTS], cnt[SPLIT_MAXPARTS], mc[SPLIT_MAXPARTS];
};
__device__ __forceinline__ int split_src(int k, const SplitPlan &P) {
  const int MO = P.mo, XO = 4 * MO + 9, NXO = 4 * MO + 4;
  const int sa = P.src[0], ma = P.mc[0], xa = 4 * ma + 9;
  // section `sec` (its offset as a function of the part's capacity mi) of logical column j
  auto colmap = [&](int j, int mul, int add, int xmul) {
    for (int p = 0; p < P.nparts; ++p)
      if (j >= P.j0[p] && j < P.j0[p] + P.cnt[p]) {
        const int mi = P.mc[p];
        return P.src[p] + (xmul >= 0 ? 4 * mi + 9 + xmul * mi : mul * mi + add) + (j - P.j0[p]);
      }
    return -1;
  };
  if (k < MO) return colmap(k, 0, 0, -1);                            // Sy(col, .)
  if (k < 2 * MO) return colmap(k - MO, 1, 0, -1);                   // Ss(., col)
  if (k == 2 * MO) return sa + 2 * ma;                               // y'y
  if (k < 3 * MO + 1) return colmap(k - (2 * MO + 1), 2, 1, -1);     // p (Wy half)
  if (k == 3 * MO + 1) return sa + 3 * ma + 1;
  if (k < 4 * MO + 2) return colmap(k - (3 * MO + 2), 3, 2, -1);     // p (Ws half)
  if (k < XO) return sa + 4 * ma + 2 + (k - (4 * MO + 2));  // new Ws column . d, f1, counts, g'd, #iwhere changed
  const int k2 = k - XO;
  if (k2 < 4 * MO) return colmap(k2 % MO, 0, 0, k2 / MO);            // the four new-row vectors
  if (k2 < NXO) return sa + xa + 4 * ma + (k2 - 4 * MO);             // their four scalars
  return sa + xa + 4 * ma + 4 + (k2 - NXO);                          // bkmin, |proj g|
}
__global__ __launch_bounds__(BLOCK) void update_split_merge_kernel(double *res, SplitPlan P, int dst) {
  const int NO = 8 * P.mo + 15;
  for (int k = threadIdx.x; k < NO; k += BLOCK) {
    const int s = split_src(k, P);
    res[dst + k] = s >= 0 ? res[s] : 0.0;
  }
}

template <typename T>
void launch_update_scan(Queue &q, int64_t n, const T *x, const T *l, const T *u, const nb_t *nbd,
                        const T *g, const T *r, const T *d, int dimpl, double stp, iw_t *iwhere,
                        T *tbrk, WStore<T> w, int head, int col, int itail, int store_pair,
                        int store_iw, int newrow, double cand_hi, uint64_t *ckeys, uint32_t *cidx,
                        uint32_t ccap, uint32_t *ccount, int ub) {
  const int nold = col - 1;
  if (newrow && nold > q.tune.split_from) {  // the split pass (see above)
    const int nparts = split_parts(nold, q.tune.split_cols);
    if (nparts > SPLIT_MAXPARTS || q.part_sel != 0 || store_pair) {
      if (q.launch_err == hipSuccess) q.launch_err = hipErrorInvalidValue, q.launch_err_where = "update_scan: split pass";
      return;
    }
    SplitPlan P{};
    P.nparts = nparts, P.mo = maxc_stride(nold);
    const int dst = q.res_off, base = std::max(split_base(nold, dst), q.split_base_min);
    int j0 = 0;
    for (int k = 0; k < nparts; ++k) {
      const int cnt = (nold - j0 + (nparts - k) - 1) / (nparts - k);  // balanced: 21 -> 11 + 10, 40 -> 14 + 13 + 13
      P.src[k] = base + k * SPLIT_SLOTS, P.j0[k] = j0, P.cnt[k] = cnt, P.mc[k] = maxc_for(cnt);
      q.res_off = P.src[k];
      launch_update_scan<T>(q, n, x, l, u, nbd, g, r, d, dimpl, stp, iwhere, k == 0 ? tbrk : (T *)nullptr, w,
                            (head - 1 + j0) % w.m + 1, cnt + 1, itail, 0, k == 0 ? store_iw : 0, 1, -1.0, ckeys,
                            cidx, ccap, ccount, ub);
      j0 += cnt;
    }
    q.res_off = dst;
    hipLaunchKernelGGL(update_split_merge_kernel, dim3(1), dim3(BLOCK), 0, q.stream, q.d_res, P, dst);
    LB_LAUNCHED(q);
    return;
  }
  if (cand_hi >= 0.0) (void)hipMemsetAsync(ccount, 0, sizeof(uint32_t), q.stream);
  int gr = 0;
  if (w.lmask) {  // W in the tile-local free-row layout
    bool done = false;
    if constexpr (sizeof(T) == 8) {
      if (nold <= 10) {
        const bool nrw = update_scan_extra(nold, newrow) != 0;
#define LB_UPDSCAN_CW(MCV, NTV, NRV)                                                                   \
  {                                                                                                    \
    gr = grid_for_w(q, n, VecOf<T>::V, (const void *)&update_scan_kernel<T, MCV, NTV, false, NRV, false, true>); \
    hipLaunchKernelGGL((update_scan_kernel<T, MCV, NTV, false, NRV, false, true>), dim3(gr), dim3(BLOCK), 0,     \
                       q.stream, n, x, l, u, nbd, g, r, d, dimpl, stp, iwhere, tbrk, w.ws, w.wy, w.zero, w.ld,   \
                       w.m, head, nold, itail, store_pair, store_iw, cand_hi, ckeys, cidx, ccap, ccount, ub,     \
                       q.part(), w.lmask);                                                                       \
  }
        int nblocks = -1;
        if (nold > 5 && nrw && q.tune.pair_cw && n >= CW_TILE) {
          // MC = 10 with the new-row sums: lane pairs share the per-column accumulators (UpdScanPairTripCW) over
          // the whole tiles; the rows behind them (< one tile) go to the plain instantiation as one more workgroup,
          // whose partials land in column `gr` of the partial-sum matrix -- as the natural-order MC = 20 pass does
          const int64_t n_main = n / CW_TILE * CW_TILE, n_rest = n - n_main;
#define LB_PAIR_CW(NTV, TIGHTV)                                                                                   \
  {                                                                                                               \
    gr = grid_for_w(q, n_main, VecOf<T>::V,                                                                       \
                    (const void *)&update_scan_kernel<T, 10, NTV, false, true, true, true, TIGHTV>);              \
    hipLaunchKernelGGL((update_scan_kernel<T, 10, NTV, false, true, true, true, TIGHTV>), dim3(gr), dim3(BLOCK), 0, \
                       q.stream, n_main, x, l, u, nbd, g, r, d, dimpl, stp, iwhere, tbrk, w.ws, w.wy, w.zero,       \
                       w.ld, w.m, head, nold, itail, store_pair, store_iw, cand_hi, ckeys, cidx, ccap, ccount, ub,   \
                       q.part(), w.lmask);                                                                          \
  }
          if (nold == 9) {
            if (q.nt) LB_PAIR_CW(true, true) else LB_PAIR_CW(false, true)
          } else {
            if (q.nt) LB_PAIR_CW(true, false) else LB_PAIR_CW(false, false)
          }
#undef LB_PAIR_CW
          nblocks = gr;
          if (n_rest > 0) {
            LB_LAUNCHED(q);
            const int64_t o = n_main;
            hipLaunchKernelGGL((update_scan_kernel<T, 10, false, false, true, false, true>), dim3(1), dim3(BLOCK), 0,
                               q.stream, n_rest, x + o, (ub & 1) ? l : l + o, (ub & 2) ? u : u + o,
                               (ub & 4) ? nbd : nbd + o, g + o, r + o, d + o, dimpl, stp, iwhere + o,
                               tbrk ? tbrk + o : tbrk, w.ws + o, w.wy + o, w.zero, w.ld, w.m, head, nold, itail,
                               store_pair, store_iw, -1.0, ckeys, cidx, ccap, ccount, ub, q.part() + gr,
                               w.lmask + o / 64);
            nblocks = gr + 1;
          }
        } else if (nold <= 5) {
          if (q.nt) { if (nrw) LB_UPDSCAN_CW(5, true, true) else LB_UPDSCAN_CW(5, true, false) }
          else { if (nrw) LB_UPDSCAN_CW(5, false, true) else LB_UPDSCAN_CW(5, false, false) }
        } else {
          if (q.nt) { if (nrw) LB_UPDSCAN_CW(10, true, true) else LB_UPDSCAN_CW(10, true, false) }
          else { if (nrw) LB_UPDSCAN_CW(10, false, true) else LB_UPDSCAN_CW(10, false, false) }
        }
#undef LB_UPDSCAN_CW
        LB_LAUNCHED(q);
        launch_finalize(q, nblocks >= 0 ? nblocks : gr, 4 * maxc_for(nold) + 9 + update_scan_extra(nold, newrow), 1, 1);
        done = true;
      }
    }
    if (!done && q.launch_err == hipSuccess)
      q.launch_err = hipErrorInvalidValue, q.launch_err_where = "update_scan: compact W needs fp64, col - 1 <= 10";
    return;
  }
#define LB_UPDSCAN(NEWROWV)                                                                          \
  DISPATCH_MAXC_NT(nold, q.nt, DISPATCH_PIPE(MC, {                                                   \
                     constexpr bool NRV = NEWROWV && MC <= 20;                                       \
                     constexpr bool PPV = MC <= 10 ? (PIPEV || NRV) : (PIPEV && !NRV && MC <= 20);   \
                     gr = grid_for_w(q, n, VecOf<T>::V,                                              \
                                     (const void *)&update_scan_kernel<T, MC, NTV, PPV, NRV>);       \
                     hipLaunchKernelGGL((update_scan_kernel<T, MC, NTV, PPV, NRV>), dim3(gr),        \
                                        dim3(BLOCK), 0, q.stream, n, x, l, u, nbd, g, r, d, dimpl,   \
                                        stp,                                                         \
                                        iwhere, tbrk, w.ws, w.wy, w.zero, w.ld, w.m, head, nold,     \
                                        itail, store_pair, store_iw, cand_hi, ckeys, cidx, ccap,     \
                                        ccount, ub, q.part());                                       \
                   }))
  const int mc = maxc_for(nold);
  // MC = 20 with the new-row sums: lane pairs share the per-column accumulators (PAIR), which
  // leaves room for the second trip in flight.  Pairs must be complete: the pair kernel takes the
  // largest multiple of 2 V rows, the plain instantiation the few rows that remain (as one more
  // workgroup: its partials go to column `gr` of the partial-sum matrix)
  // Measured (n = 5e7 / 1e8, m = 20), two trips in flight: fp64 3.04 -> 2.78 ms, fp32 3.55 -> 3.17 ms with round 3's
  // form of the kernel (every lane loaded all columns, halves exchanged by DPP: 509 registers in fp32, where one
  // spilled register's scratch reloads, returning in order behind the loads in flight, undid the pipelining: 4.68
  // ms); with the column halves loaded per lane (UpdScanPairTrip: 450 registers) fp64 2.51 - 2.53, fp32 2.56 ms
  // Tune::pair = 0: off; 1: on, one trip in flight; 2: two trips
  const int pair_mode = q.tune.pair;
  int nblocks = 0;
  if (update_scan_extra(nold, newrow) && mc == 20 && pair_mode > 0) {
    constexpr int MC = 20;
    gr = grid_for_w(q, n, VecOf<T>::V, (const void *)&update_scan_kernel<T, MC, true, true, true, true>);
    nblocks = gr;
    constexpr int VP = RowsPerAcc<T, MC, 4 * MC + 11 + 4 * MC + 4>::V;
    const int64_t n_main = n / (2 * VP) * (2 * VP), n_rest = n - n_main;
#define LB_PAIR(NTV, PIPEV)                                                                          \
  hipLaunchKernelGGL((update_scan_kernel<T, MC, NTV, PIPEV, true, true>), dim3(gr), dim3(BLOCK), 0,  \
                     q.stream, n_main, x, l, u, nbd, g, r, d, dimpl, stp, iwhere, tbrk, w.ws, w.wy,  \
                     w.zero, w.ld, w.m, head, nold, itail, store_pair, store_iw, -1.0, ckeys, cidx,   \
                     ccap, ccount, ub, q.part())
    if (n_main > 0) {
      if (q.nt) {
        if (pair_mode >= 2) LB_PAIR(true, true); else LB_PAIR(true, false);
      } else {
        if (pair_mode >= 2) LB_PAIR(false, true); else LB_PAIR(false, false);
      }
    }
#undef LB_PAIR
    if (n_rest > 0 || n_main == 0) {
      const int64_t o = n_main;
      hipLaunchKernelGGL((update_scan_kernel<T, MC, false, false, true>), dim3(1), dim3(BLOCK), 0, q.stream,
                         n_rest, x + o, (ub & 1) ? l : l + o, (ub & 2) ? u : u + o, (ub & 4) ? nbd : nbd + o,
                         g + o, r + o, d + o, dimpl, stp, iwhere + o,
                         tbrk ? tbrk + o : tbrk, w.ws + o, w.wy + o, w.zero, w.ld, w.m, head, nold, itail,
                         store_pair, store_iw, -1.0, ckeys, cidx, ccap, ccount, ub,
                         q.part() + (n_main > 0 ? gr : 0));
      nblocks = n_main > 0 ? gr + 1 : 1;
      LB_LAUNCHED(q);
    }
  } else if (update_scan_extra(nold, newrow)) {
    LB_UPDSCAN(true);
    nblocks = gr;
  } else {
    LB_UPDSCAN(false);
    nblocks = gr;
  }
#undef LB_UPDSCAN
  LB_LAUNCHED(q);
  launch_finalize(q, nblocks, 4 * mc + 9 + update_scan_extra(nold, newrow), 1, 1);
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_update_pairs<T>(Queue &, int64_t, const T *, const T *, const T *, double, WStore<T>, int, int, int); \
  template void launch_update_scan<T>(Queue &, int64_t, const T *, const T *, const T *, const nb_t *, const T *, const T *, const T *, int, double, iw_t *, T *, WStore<T>, int, int, int, int, int, int, double, uint64_t *, uint32_t *, uint32_t, uint32_t *, int);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
