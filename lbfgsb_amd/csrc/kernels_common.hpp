// kernels_common.hpp / k_*.hip -- hand-written gfx950 (CDNA4) kernels for the n-dimensional work
// of the L-BFGS-B iteration (reference src/lbfgsb.f90, routines cited per kernel).
//
// Shape of every kernel: tall-skinny, HBM-bound, no reuse.  One lane owns V
// consecutive rows (16 B per array per load, dwordx4), grid-stride over rows,
// <= 2048 workgroups of 4 wave64 (768 for the passes over W: what is resident).
// The correction-pair matrices Ws, Wy are column-major with a 256-byte aligned
// leading dimension, so lane i of a wave reads 16 B at column_base + 16*i: every
// wave-instruction is one fully coalesced 1 KiB request per column.  Reductions:
// per-lane fp64 accumulators -> wave shuffle -> LDS across the 4 waves -> one
// partial per workgroup -> fixed-order finalize kernel (deterministic; no float
// atomics).  2m <= 64 columns is far too thin for MFMA: the flop/byte ratio is
// <= 2 (fp64), the machine balance ~10, so the roofline is HBM bandwidth.
//
// Column loops are unrolled to a compile-time MAXC; logical columns >= col load a
// 256-byte zero buffer (WStore::zero: an L1/L2 hit, no HBM traffic) and contribute
// nothing, which keeps every load unconditional and in flight together.  The two
// passes of the steady-state iteration (k_update.hip, k_subsm.hip) and the
// three-pass fallback (k_cmprlb.hip) schedule their loads themselves: for_rows_raw
// in device_util.hpp, DESIGN.md section 4.
//
// kernels_common.hpp -- pieces shared by the kernel translation units (k_*.hip): launch-size
// helpers, the column-count dispatch macros, the circular column addressing of W, the pending
// pair, and small per-row formulas that several kernels must evaluate identically.
#pragma once
#include "kernels.hpp"

#include <cstdlib>
#include <cstring>

#include "device_util.hpp"

namespace lbk {

#define LB_INF (__builtin_huge_val())

// the same, also selecting the load policy at run time (q.nt)
#define DISPATCH_MAXC_NT(col, ntflag, ...)   \
  do {                                        \
    if (ntflag) {                             \
      constexpr bool NTV = true;              \
      DISPATCH_MAXC(col, __VA_ARGS__);        \
    } else {                                  \
      constexpr bool NTV = false;             \
      DISPATCH_MAXC(col, __VA_ARGS__);        \
    }                                         \
  } while (0)

#define DISPATCH_MAXC(col, ...)       \
  do {                                \
    if ((col) <= 5) {                 \
      constexpr int MC = 5;           \
      __VA_ARGS__;                    \
    } else if ((col) <= 10) {         \
      constexpr int MC = 10;          \
      __VA_ARGS__;                    \
    } else if ((col) <= 20) {         \
      constexpr int MC = 20;          \
      __VA_ARGS__;                    \
    } else {                          \
      constexpr int MC = 32;          \
      __VA_ARGS__;                    \
    }                                 \
  } while (0)

// Two trips in flight per wave (for_rows_raw, PIPE) where a kernel's registers leave one wave
// per SIMD anyway: the MC = 20 instantiations (MC = 32 would need a vmcnt beyond 63).
// Measured (bench.py, n = 1e8): fp32 m = 20 89.7 -> 96.2 it/s, fp32 m = 10 159.5 -> 163.3, fp64
// m = 10 no change (its kernels already run 2-3 waves per SIMD): on for MC = 20 and for fp32 MC = 10.
// Only those shapes are COMPILED with PIPE (the others spilled to scratch with two register sets and
// were reachable through the option alone); Tune::pipe = 0 switches it off, for A/B timings.
bool pipe_on(const Queue &q, int mc, int elem_bytes);  // k_misc.hip
#define DISPATCH_PIPE(MCV, ...)                                        \
  do {                                                                 \
    if constexpr ((MCV) == 20 || ((MCV) == 10 && sizeof(T) == 4)) {    \
      if (pipe_on(q, MCV, (int)sizeof(T))) {                           \
        constexpr bool PIPEV = true;                                   \
        __VA_ARGS__;                                                   \
      } else {                                                         \
        constexpr bool PIPEV = false;                                  \
        __VA_ARGS__;                                                   \
      }                                                                \
    } else {                                                           \
      constexpr bool PIPEV = false;                                    \
      __VA_ARGS__;                                                     \
    }                                                                  \
  } while (0)

// physical column offset (elements) of logical column j; j >= col -> logical 0
__device__ __forceinline__ int64_t col_off(int j, int col, int head, int m, int64_t ld) {
  const int jj = j < col ? j : 0;
  return (int64_t)((head - 1 + jj) % m) * ld;
}
// Unroll slots beyond the stored pairs (the kernels are unrolled to MC = 5/10/20/32 columns;
// e.g. the update pass at col - 1 = 9 old columns runs the MC = 10 code) read a small zero
// buffer instead (`zero`: the same 16 bytes for every lane -- a cache hit, no HBM traffic).
// Selecting the ADDRESS keeps the load unconditional: a branch around each load cuts the loop
// body into basic blocks, the compiler then waits for the first loads of a trip (s_waitcnt
// vmcnt(0) behind the iwhere unpack) before it has issued the column loads, and a wave that runs
// alone on its SIMD (fp32, m = 20: > 256 registers) pays two memory latencies per trip -- round
// 1's 3.7 TB/s for that kernel against 6.4 TB/s for the same arithmetic in straight-line form
// (profiles/scripts/r32m20_variants.hip).
template <typename T, int W, bool NT>
__device__ __forceinline__ void ld_col(bool live, const T *p, const T *zero, double (&o)[W]) {
  ldx<W, NT>(live ? p : zero, o);
}

// ---- tile-local free-row layout of W (WStore::lmask) ----
// where row `row` of a column sits: inside its aligned tile of CW_TILE = 128 rows the rows whose layout bit is set
// come first, in ascending order, the others follow in ascending order (lmask == nullptr: natural order).
// For the kernels that touch single rows (record gathers, formk's patch); the two passes over W derive the slots of
// a whole tile from its two mask words (for_tiles_cw).
__device__ __forceinline__ int64_t wrow(const uint64_t *__restrict__ lmask, int64_t row) {
  if (!lmask) return row;
  const int64_t tile = row >> 7;
  const int r = (int)(row & 127);
  const uint64_t m0 = lmask[2 * tile], m1 = lmask[2 * tile + 1];
  const int c0 = __popcll(m0), tf = c0 + __popcll(m1);
  const bool hi = r >= 64;
  const uint64_t mw = hi ? m1 : m0;
  const int b = r & 63;
  const int before = (hi ? c0 : 0) + __popcll(mw & ((1ull << b) - 1ull));
  const bool fr = (mw >> b) & 1ull;
  return (tile << 7) + (fr ? before : tf + (r - before));
}

// Where a trip's row group lives -- mixed into the trip types of the two passes over W, so that ONE kernel body
// serves the natural order (W consecutive rows from i) and the tile-local layout (CwRows / one row and its slot).
template <int W>
struct NaturalRows {
  static constexpr bool CW = false;
  __device__ __forceinline__ int64_t row(int64_t i, int k) const { return i + k; }
  template <bool NTS, typename E>
  __device__ __forceinline__ void st_rows(E *p, int64_t i, const double (&v)[W]) const {  // p[row k] = v[k]
    if constexpr (NTS) stnt<W>(p + i, v); else st<W>(p + i, v);
  }
  template <bool NTS, typename E>
  __device__ __forceinline__ void st_w(E *col, int64_t i, const double (&v)[W]) const {  // into a W column
    st_rows<NTS, E>(col, i, v);
  }
  __device__ __forceinline__ void sti_rows(iw_t *p, int64_t i, const int (&v)[W]) const { sti<W>(p + i, v); }
};
struct CwPairRows {  // the rows lane, lane + 64 of a full tile (for_tiles_cw): derived from the tile's scalars on demand
  static constexpr bool CW = true;
  CwTile tile_;
  __device__ __forceinline__ int64_t row(int64_t, int k) const { return tile_.tb + (int64_t)((threadIdx.x & 63) + 64 * k); }
  __device__ __forceinline__ bool lf(int k) const { return ((k ? tile_.m1 : tile_.m0) >> (threadIdx.x & 63)) & 1ull; }
  template <bool NTS, typename E>
  __device__ __forceinline__ void st_rows(E *p, int64_t, const double (&v)[2]) const {
    E *q = p + tile_.tb;  // (uniform)
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double one[1] = {v[k]};
      if constexpr (NTS) stnt<1>(q + (lane + 64 * k), one); else st<1>(q + (lane + 64 * k), one);
    }
  }
  template <bool NTS, typename E>
  __device__ __forceinline__ void st_w(E *col, int64_t, const double (&v)[2]) const {
    int sl[2];
    bool f[2];
    cw_slots(tile_, sl, f);
    E *q = col + tile_.tb;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double one[1] = {v[k]};
      if constexpr (NTS) stnt<1>(q + sl[k], one); else st<1>(q + sl[k], one);
    }
  }
  __device__ __forceinline__ void sti_rows(iw_t *p, int64_t, const int (&v)[2]) const {
    iw_t *q = p + tile_.tb;
    const int lane = (int)(threadIdx.x & 63);
    q[lane] = (iw_t)v[0], q[lane + 64] = (iw_t)v[1];
  }
};
struct CwOneRow {  // one row of the partial tile
  static constexpr bool CW = true;
  int64_t ri_, ws_;
  bool lf_;
  __device__ __forceinline__ int64_t row(int64_t, int) const { return ri_; }
  __device__ __forceinline__ bool lf(int) const { return lf_; }
  template <bool NTS, typename E>
  __device__ __forceinline__ void st_rows(E *p, int64_t, const double (&v)[1]) const {
    if constexpr (NTS) stnt<1>(p + ri_, v); else st<1>(p + ri_, v);
  }
  template <bool NTS, typename E>
  __device__ __forceinline__ void st_w(E *col, int64_t, const double (&v)[1]) const {
    if constexpr (NTS) stnt<1>(col + ws_, v); else st<1>(col + ws_, v);
  }
  __device__ __forceinline__ void sti_rows(iw_t *p, int64_t, const int (&v)[1]) const { p[ri_] = (iw_t)v[0]; }
};
// two single bytes -> the register image of a 2-byte load (raw_geti<2>)
__device__ __forceinline__ void raw_join_bytes(RawReg<2> &r, const RawReg<1> &a, const RawReg<1> &b) {
  r.v = (a.v & 0xff) | ((b.v & 0xff) << 8);
}

// ---- pending pair ----
// Between matupd and the subspace pass of the same setulb call the newest pair (logical column
// col-1) is not in W yet: update_scan_kernel only reduces, so that it stays a read-only pass
// (a single store stream drops a streaming pass on MI355X from ~6.5 to ~4.8 TB/s,
// profiles/scripts/write_cost.hip).  Until subsm_update_kernel -- which stores vectors anyway --
// commits it, the column is defined by the vectors it was formed from, with the rounding of a
// store to T:   y = T(g - r),   s = T(stp * d)   (mainlb :813-822, matupd :2313-2314).
template <typename T>
__device__ __forceinline__ double pend_y(double gk, double rk) {
  return (double)(T)(gk - rk);
}
template <typename T>
__device__ __forceinline__ double pend_s(double dk, double stp) {
  return stp != 1.0 ? (double)(T)(stp * dk) : dk;
}
// the same when d is implicit (Pend::impl): `tk` is what was loaded through pd (= t)
template <typename T>
__device__ __forceinline__ double pend_sx(double tk, double xk, Pend pe) {
  return pe.impl ? (double)(T)(xk - tk) : pend_s<T>(tk, pe.stp);
}
// columns j = 0..MC-1 of one row group; the pending column is read from (r, d) instead.
// PSPEC: the caller guarantees pe.on && col == MC (every iteration once the memory is full):
// the pending column is then the compile-time slot MC - 1 and no select is needed.
template <typename T, int MC, int W, bool NT, bool PSPEC = false>
__device__ __forceinline__ void load_cols(const T *__restrict__ wy, const T *__restrict__ ws,
                                          const T *pr, const T *pd, const T *zero, int64_t i,
                                          int col, int head, int m, int64_t ldw, Pend pe,
                                          double (&a)[MC][W], double (&b)[MC][W]) {
  if constexpr (PSPEC) {
#pragma unroll
    for (int j = 0; j < MC - 1; ++j) {
      const int64_t off = (int64_t)((head - 1 + j) % m) * ldw + i;
      ldx<W, NT>(wy + off, a[j]);
      ldx<W, NT>(ws + off, b[j]);
    }
    ldx<W, NT>(pr + i, a[MC - 1]);
    ldx<W, NT>(pd + i, b[MC - 1]);
  } else {
    // one base pointer per matrix and a selected element offset (selecting between two base
    // pointers per column makes the compiler keep a table of addresses in scratch memory);
    // all buffers are allocations of T, so the distances are whole elements
    const int64_t dy = pe.on ? (int64_t)(((intptr_t)pr - (intptr_t)wy) / (intptr_t)sizeof(T)) : 0;
    const int64_t ds = pe.on ? (int64_t)(((intptr_t)pd - (intptr_t)ws) / (intptr_t)sizeof(T)) : 0;
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, col, head, m, ldw);
      const bool pj = pe.on && j == col - 1;
      ld_col<T, W, NT>(j < col, wy + ((pj ? dy : off) + i), zero, a[j]);
      ld_col<T, W, NT>(j < col, ws + ((pj ? ds : off) + i), zero, b[j]);
    }
  }
}
template <typename T, int MC, int W, bool PSPEC = false>
__device__ __forceinline__ void fix_pending(int col, Pend pe, const double (&gv)[W],
                                            const double (&xv)[W], double (&a)[MC][W],
                                            double (&b)[MC][W]) {
  if constexpr (PSPEC) {
#pragma unroll
    for (int k = 0; k < W; ++k) {
      a[MC - 1][k] = pend_y<T>(gv[k], a[MC - 1][k]);
      b[MC - 1][k] = pend_sx<T>(b[MC - 1][k], xv[k], pe);
    }
  } else {
    // branch-free selects: a predicated write a[col-1][k] = ... would turn the register arrays
    // into dynamically indexed ones (scratch memory)
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const bool pj = pe.on && j == col - 1;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const double yk = pend_y<T>(gv[k], a[j][k]);
        const double sk = pend_sx<T>(b[j][k], xv[k], pe);
        a[j][k] = pj ? yk : a[j][k];
        b[j][k] = pj ? sk : b[j][k];
      }
    }
  }
}
// The same in two phases for the kernels that schedule their loads themselves (device_util.hpp,
// "raw" loads): issue_cols starts every column load of the trip, get_cols reads the landed
// registers.
template <typename T, int MC, int W, bool NT, bool PSPEC = false>
__device__ __forceinline__ void issue_cols(const T *__restrict__ wy, const T *__restrict__ ws,
                                           const T *pr, const T *pd, const T *zero, int64_t i,
                                           int col, int head, int m, int64_t ldw, Pend pe,
                                           RawOf<T, W> (&ra)[MC], RawOf<T, W> (&rb)[MC]) {
  constexpr int B = (int)sizeof(T) * W;
  if constexpr (PSPEC) {
#pragma unroll
    for (int j = 0; j < MC - 1; ++j) {
      const int64_t off = (int64_t)((head - 1 + j) % m) * ldw + i;
      raw_issue<B, NT>(ra[j], wy + off);
      raw_issue<B, NT>(rb[j], ws + off);
    }
    raw_issue<B, NT>(ra[MC - 1], pr + i);
    raw_issue<B, NT>(rb[MC - 1], pd + i);
  } else {
    // (one base pointer per matrix and a selected element offset, as in load_cols)
    const int64_t dy = pe.on ? (int64_t)(((intptr_t)pr - (intptr_t)wy) / (intptr_t)sizeof(T)) : 0;
    const int64_t ds = pe.on ? (int64_t)(((intptr_t)pd - (intptr_t)ws) / (intptr_t)sizeof(T)) : 0;
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, col, head, m, ldw);
      const bool live = j < col, pj = pe.on && j == col - 1;
      const T *py = wy + ((pj ? dy : off) + i), *ps = ws + ((pj ? ds : off) + i);
      raw_issue<B, NT>(ra[j], live ? py : zero);
      raw_issue<B, NT>(rb[j], live ? ps : zero);
    }
  }
}
template <typename T, int MC, int W>
__device__ __forceinline__ void land_cols(RawOf<T, W> (&ra)[MC], RawOf<T, W> (&rb)[MC]) {
#pragma unroll
  for (int j = 0; j < MC; ++j) {
    raw_land(ra[j]);
    raw_land(rb[j]);
  }
}
template <typename T, int MC, int W>
__device__ __forceinline__ void get_cols(const RawOf<T, W> (&ra)[MC], const RawOf<T, W> (&rb)[MC],
                                         double (&a)[MC][W], double (&b)[MC][W]) {
#pragma unroll
  for (int j = 0; j < MC; ++j) {
    raw_get<W>(ra[j], (const T *)nullptr, a[j]);
    raw_get<W>(rb[j], (const T *)nullptr, b[j]);
  }
}
// One column pair (logical column j) of a landed trip as doubles, the pending pair already in
// its stored form.  For the kernels that must not keep all 2*MC operands widened at once (fp32
// with many accumulators): they read each column where they use it, twice if need be -- OPAQUE
// hides the second conversion from the optimiser, which would otherwise keep the doubles of the
// first use alive (and spill).
__device__ __forceinline__ double cvt_opaque(double v) { return v; }
__device__ __forceinline__ double cvt_opaque(float v) {
  double d;
  asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(v));
  return d;
}
template <int W, bool OPAQUE>
__device__ __forceinline__ void raw_get_col(const RawReg<8 * W> &r, const double *, double (&o)[W]) {
  raw_get<W>(r, (const double *)nullptr, o);
}
template <int W, bool OPAQUE>
__device__ __forceinline__ void raw_get_col(const RawReg<4 * W> &r, const float *, double (&o)[W]) {
  if constexpr (!OPAQUE) {
    raw_get<W>(r, (const float *)nullptr, o);
  } else if constexpr (W == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_bit_cast(f4, r.v);
    o[0] = cvt_opaque(v.x), o[1] = cvt_opaque(v.y), o[2] = cvt_opaque(v.z), o[3] = cvt_opaque(v.w);
  } else if constexpr (W == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = __builtin_bit_cast(f2, r.v);
    o[0] = cvt_opaque(v.x), o[1] = cvt_opaque(v.y);
  } else {
    o[0] = cvt_opaque(__builtin_bit_cast(float, r.v));
  }
}
template <typename T, int MC, int W, bool PSPEC, bool OPAQUE>
__device__ __forceinline__ void col_pair(const RawOf<T, W> (&ra)[MC], const RawOf<T, W> (&rb)[MC], int j,
                                         int col, Pend pe, const double (&gv)[W], double (&aj)[W],
                                         double (&bj)[W]) {
  raw_get_col<W, OPAQUE>(ra[j], (const T *)nullptr, aj);
  raw_get_col<W, OPAQUE>(rb[j], (const T *)nullptr, bj);
  const bool pj = PSPEC ? j == MC - 1 : (pe.on && j == col - 1);
  if (PSPEC ? j == MC - 1 : true) {  // (compile-time false for the other columns of PSPEC)
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const double yk = pend_y<T>(gv[k], aj[k]);
      const double sk = pend_s<T>(bj[k], pe.stp);
      aj[k] = pj ? yk : aj[k];
      bj[k] = pj ? sk : bj[k];
    }
  }
}

// the newest pair (logical column col - 1) of one row group, from the registers
template <int MC, int W, bool PSPEC = false>
__device__ __forceinline__ void newest_cols(int col, const double (&a)[MC][W], const double (&b)[MC][W],
                                            double (&yn)[W], double (&sn)[W]) {
#pragma unroll
  for (int k = 0; k < W; ++k) {
    if constexpr (PSPEC) {
      yn[k] = a[MC - 1][k], sn[k] = b[MC - 1][k];
    } else {
      yn[k] = 0.0, sn[k] = 0.0;
#pragma unroll
      for (int j = 0; j < MC; ++j)
        if (j == col - 1) {
          yn[k] = a[j][k];
          sn[k] = b[j][k];
        }
    }
  }
}

// fixed-order reduction of per-block partials (k_misc.hip)
void finalize_from(Queue &q, const double *part, int pstride, int nblocks, int nsum, int nmin,
                   int nmax);

// ---- dictionary-coded bounds (ub bit 3, UB_DICT) ----
// Bound arrays with FEW distinct values (the box [a, b]^n with some variables on another box, driver3's
// alternating l = 1 / -100, test/driver3.f90:102-120) are not streamed by the passes over W either: the
// packed one-byte copy of nbd the passes read anyway then carries  nbd | l-index << 2 | u-index << 5 , and
// the values come from two tables of <= 8 entries each -- the SAME 64-byte buffers the uniform case reads
// (l at +0, u at +64: a uniform array is the dictionary whose entries are all that one value), copied into
// LDS once per workgroup.  The table entries are the caller's values bit for bit, so the arithmetic and
// every result are those of the streaming kernels; 1 byte per row instead of 2 x sizeof(T) + 1.
template <typename T>
__device__ __forceinline__ void dict_fill(T (&tab)[16], const T *l, const T *u, int ub) {
  if (ub & UB_DICT) {  // (uniform over the grid)
    if (threadIdx.x < 16) tab[threadIdx.x] = threadIdx.x < 8 ? l[threadIdx.x] : u[threadIdx.x - 8];
    __syncthreads();
  }
}
template <typename T, int W>
__device__ __forceinline__ void dict_apply(const T (&tab)[16], int ub, int (&nb)[W], double (&lv)[W],
                                           double (&uv)[W]) {
  if (ub & UB_DICT) {
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const unsigned c = (unsigned)nb[k] & 0xffu;
      lv[k] = (double)tab[(c >> 2) & 7u];
      uv[k] = (double)tab[8u + (c >> 5)];
      nb[k] = (int)(c & 3u);
    }
  }
}

// ---- projgr (:2594-2622) of one row ----
__device__ __forceinline__ double proj_g(double x, double l, double u, int nb, double gi) {
  if (nb != 0) {
    if (gi < 0.0) {
      if (nb >= 2) gi = fmax(x - u, gi);
    } else {
      if (nb <= 2) gi = fmin(x - l, gi);
    }
  }
  return fabs(gi);
}

// ---- ordering of breakpoints ----
__device__ __forceinline__ bool after_cursor(double t, int64_t gi, double lo_t, int64_t lo_i) {
  return t > lo_t || (t == lo_t && gi > lo_i);
}
__device__ __forceinline__ uint64_t key_of(double t) {  // t >= 0: bit pattern is monotone
  return (uint64_t)__double_as_longlong(t);
}

// Breakpoint time of one row from its own data, exactly as the scans store it in tbrk (incl.
// the rounding to T): -1 = the row does not move, +inf = it moves without meeting a bound.
// iw is iwhere AFTER the scan's update (cauchy :1284-1291).
template <typename T>
__device__ __forceinline__ double brk_time(double xk, double lk, double uk, int nb, double gk,
                                           int iw) {
  if (iw != 0 && iw != -1) return -1.0;
  const double neggi = -gk;
  double tb = LB_INF;
  if (nb <= 2 && nb != 0 && neggi < 0.0) {
    tb = (xk - lk) / (-neggi);
  } else if (nb >= 2 && neggi > 0.0) {
    tb = (uk - xk) / neggi;
  }
  return (double)(T)tb;
}

// The generalized Cauchy point is not stored as a vector on the main path: after the walk,
// xcp(k) is a function of row k's own x, g, bounds and iwhere (cauchy :1341, :1425-1433, :1515):
//   iwhere in {0,-1} (the row moves with d = -g and was not fixed):  x + tsum*d
//   iwhere == 1 / 2: the bound if the WALK fixed the row there (it started strictly inside: x > l,
//     x < u), x itself if the scan found it at the bound (:1284-1291: x <= l, x >= u -- a line-search
//     step of stpmx can leave x an ulp OUTSIDE the box, and the reference does not move it back)
//   otherwise (always fixed): x
// ... each of the non-moving cases + tsum * 0: the reference ends with xcp = xcp + tsum * d over ALL
// rows (:1515) with d = 0 there, which changes nothing unless tsum is not finite -- d == 0 over all rows
// with a projected gradient of a few ulps (x outside the box by an ulp) gives dtm = 0/0, tsum = NaN, and
// the reference's Cauchy point is then NaN in EVERY component; so is this one.
// Every consumer evaluates exactly the expression cauchy_finish_kernel stores (incl. the
// rounding to T), so results do not depend on whether z was materialised.
template <typename T>
__device__ __forceinline__ double xcp_free(double xk, double gk, int iw, double tsum) {
  if (iw == 0 || iw == -1) return tsum != 0.0 ? (double)(T)(xk + tsum * (-gk)) : xk;
  return xk + tsum * 0.0;
}
template <typename T>
__device__ __forceinline__ double xcp_row(double xk, double gk, int iw, double lk, double uk,
                                          double tsum) {
  if (tsum == 0.0) return xk;  // a walk that fixed a row has tsum >= its breakpoint > 0
  if (iw == 1) return (xk > lk ? lk : xk) + tsum * 0.0;
  if (iw == 2) return (xk < uk ? uk : xk) + tsum * 0.0;
  return xcp_free<T>(xk, gk, iw, tsum);
}

}  // namespace lbk
