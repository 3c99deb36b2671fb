// device_util.hpp -- wave64 / workgroup building blocks for the streaming kernels.
//
// gfx950 (CDNA4): 64-lane wavefronts, 4 waves per 256-thread workgroup.  All
// kernels here are HBM-bound tall-skinny streams: each lane issues 16-byte
// loads (dwordx4) straight into VGPRs -- the operands are read once and not
// shared between waves, so an LDS round trip would be pure overhead (cdna
// guide, "GEMV / M <= 16" row) -- and LDS is used only for the cross-wave step
// of reductions and for the row tiles of the formk Gram kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace lbk {

template <typename T>
struct VecOf;
template <>
struct VecOf<double> {
  static constexpr int V = 2;  // 2 x f64 = 16 B per lane
};
template <>
struct VecOf<float> {
  static constexpr int V = 4;  // 4 x f32 = 16 B per lane
};

template <int W>
using WTag = std::integral_constant<int, W>;

// ---- W consecutive elements, as doubles ----
template <int W>
__device__ __forceinline__ void ld(const double *p, double (&o)[W]) {
  if constexpr (W == 2) {
    double2 v = *reinterpret_cast<const double2 *>(p);
    o[0] = v.x;
    o[1] = v.y;
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) o[k] = p[k];
  }
}
template <int W>
__device__ __forceinline__ void ld(const float *p, double (&o)[W]) {
  if constexpr (W == 2) {
    float2 v = *reinterpret_cast<const float2 *>(p);
    o[0] = v.x;
    o[1] = v.y;
  } else if constexpr (W == 4) {
    float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
    o[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) o[k] = p[k];
  }
}
// ---- loads with a cache policy: NT = nontemporal (streamed once; on MI355X the W'v stream
//      reads 6.55 TB/s with nt loads against 5.89 TB/s with plain ones, profiles/scripts/
//      wtv_variants.hip).  Plain loads are kept for problems whose W fits the 256 MiB
//      Infinity Cache, where the next kernel re-reads it on-die. ----
template <int W, bool NT>
__device__ __forceinline__ void ldx(const double *p, double (&o)[W]) {
  if constexpr (NT && W == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
    o[0] = v.x;
    o[1] = v.y;
  } else if constexpr (NT) {
#pragma unroll
    for (int k = 0; k < W; ++k) o[k] = __builtin_nontemporal_load(p + k);
  } else {
    ld<W>(p, o);
  }
}
template <int W, bool NT>
__device__ __forceinline__ void ldx(const float *p, double (&o)[W]) {
  if constexpr (NT && W == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
    o[3] = v.w;
  } else if constexpr (NT && W == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p));
    o[0] = v.x;
    o[1] = v.y;
  } else if constexpr (NT) {
#pragma unroll
    for (int k = 0; k < W; ++k) o[k] = __builtin_nontemporal_load(p + k);
  } else {
    ld<W>(p, o);
  }
}

template <int W>
__device__ __forceinline__ void st(double *p, const double (&o)[W]) {
  if constexpr (W == 2) {
    *reinterpret_cast<double2 *>(p) = make_double2(o[0], o[1]);
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) p[k] = o[k];
  }
}
template <int W>
__device__ __forceinline__ void st(float *p, const double (&o)[W]) {
  if constexpr (W == 2) {
    *reinterpret_cast<float2 *>(p) = make_float2((float)o[0], (float)o[1]);
  } else if constexpr (W == 4) {
    *reinterpret_cast<float4 *>(p) = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) p[k] = (float)o[k];
  }
}
// nontemporal stores (streamed out, not re-read by this kernel)
template <int W>
__device__ __forceinline__ void stnt(double *p, const double (&o)[W]) {
  if constexpr (W == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v = {o[0], o[1]};
    __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p));
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) __builtin_nontemporal_store(o[k], p + k);
  }
}
template <int W>
__device__ __forceinline__ void stnt(float *p, const double (&o)[W]) {
  if constexpr (W == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
  } else if constexpr (W == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {(float)o[0], (float)o[1]};
    __builtin_nontemporal_store(v, reinterpret_cast<f2 *>(p));
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) __builtin_nontemporal_store((float)o[k], p + k);
  }
}
template <int W>
__device__ __forceinline__ void ldi(const int32_t *p, int (&o)[W]) {
  if constexpr (W == 2) {
    int2 v = *reinterpret_cast<const int2 *>(p);
    o[0] = v.x;
    o[1] = v.y;
  } else if constexpr (W == 4) {
    int4 v = *reinterpret_cast<const int4 *>(p);
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
    o[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) o[k] = p[k];
  }
}
// iwhere (values -3..3) and the solver's copy of nbd (0..3) are kept as one byte per row
template <int W>
__device__ __forceinline__ void ldi(const int8_t *p, int (&o)[W]) {
  if constexpr (W == 2) {
    const char2 v = *reinterpret_cast<const char2 *>(p);
    o[0] = v.x, o[1] = v.y;
  } else if constexpr (W == 4) {
    const char4 v = *reinterpret_cast<const char4 *>(p);
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) o[k] = p[k];
  }
}
template <int W>
__device__ __forceinline__ void sti(int8_t *p, const int (&o)[W]) {
  if constexpr (W == 2) {
    *reinterpret_cast<char2 *>(p) = make_char2((signed char)o[0], (signed char)o[1]);
  } else if constexpr (W == 4) {
    *reinterpret_cast<char4 *>(p) =
        make_char4((signed char)o[0], (signed char)o[1], (signed char)o[2], (signed char)o[3]);
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) p[k] = (int8_t)o[k];
  }
}
template <int W>
__device__ __forceinline__ void sti(int32_t *p, const int (&o)[W]) {
  if constexpr (W == 2) {
    *reinterpret_cast<int2 *>(p) = make_int2(o[0], o[1]);
  } else if constexpr (W == 4) {
    *reinterpret_cast<int4 *>(p) = make_int4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) p[k] = o[k];
  }
}

// ---- loads the kernel schedules itself ("raw" loads) -------------------------------------
// The passes over W want ALL loads of a trip (2*MC columns + the n-vectors) in flight before the
// first use.  hipcc does not do that by itself when the loop body is cut into basic blocks
// (guards around single loads) or when every loaded fp32 value is widened at once (fp32, m = 20:
// load, s_waitcnt vmcnt(0), convert, next load -- a wave that runs alone on its SIMD then pays
// one memory latency per COLUMN; round 1's 3.7 TB/s).  So the kernels (a) load into register
// images of the STORAGE type (RawReg: nothing to convert, nothing to wait for), all of them in
// one straight-line run, (b) put a scheduling barrier behind the run (raw_wait: no instruction
// crosses it), and (c) convert where the value is used (raw_get).  These are ordinary loads: the
// compiler tracks them and inserts every s_waitcnt itself (counted waits across the pipelined
// trips of for_rows_raw included) -- an inline-asm load would be faster to pin down but is
// unsafe here: with more than 256 live registers the register allocator moves an asm load's
// destination to the accumulator file BEFORE the data has landed.
// Protocol, per trip:  raw_issue(...) x k;  raw_wait<N>();  raw_land(...) x k;  raw_get(...).
typedef int i32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int BYTES>
struct RawReg;
template <>
struct RawReg<16> {
  i32x4_t v;
};
template <>
struct RawReg<8> {
  i32x2_t v;
};
template <>
struct RawReg<4> {
  int v;
};
template <>
struct RawReg<2> {
  int v;  // two bytes, zero-extended
};
template <>
struct RawReg<1> {
  int v;  // one byte, sign-extended
};
template <int BYTES, bool NT>
__device__ __forceinline__ void raw_issue(RawReg<BYTES> &r, const void *p) {
  if constexpr (BYTES == 16) {
    if constexpr (NT)
      r.v = __builtin_nontemporal_load(reinterpret_cast<const i32x4_t *>(p));
    else
      r.v = *reinterpret_cast<const i32x4_t *>(p);
  } else if constexpr (BYTES == 8) {
    if constexpr (NT)
      r.v = __builtin_nontemporal_load(reinterpret_cast<const i32x2_t *>(p));
    else
      r.v = *reinterpret_cast<const i32x2_t *>(p);
  } else if constexpr (BYTES == 4) {
    if constexpr (NT)
      r.v = __builtin_nontemporal_load(reinterpret_cast<const int *>(p));
    else
      r.v = *reinterpret_cast<const int *>(p);
  } else if constexpr (BYTES == 2) {
    r.v = (int)*reinterpret_cast<const unsigned short *>(p);
  } else {
    r.v = (int)*reinterpret_cast<const signed char *>(p);
  }
}
// one 8-byte (fp64) / 4-byte (fp32) half of a two-element register image (the tile-local layout of W: a row
// group's two rows are not neighbours in memory)
template <bool NT>
__device__ __forceinline__ void raw_issue_half(RawReg<16> &r, int half, const void *p) {
  i32x2_t v;
  if constexpr (NT)
    v = __builtin_nontemporal_load(reinterpret_cast<const i32x2_t *>(p));
  else
    v = *reinterpret_cast<const i32x2_t *>(p);
  if (half == 0)
    r.v.xy = v;
  else
    r.v.zw = v;
}
// ... from a wave-uniform base pointer + a per-lane byte offset < 2^32 (scalar base + 32-bit vector offset: no
// 64-bit address arithmetic per lane and load)
template <bool NT>
__device__ __forceinline__ void raw_issue_half_at(RawReg<16> &r, int half, const void *base, uint32_t byte_off) {
  raw_issue_half<NT>(r, half, (const char *)base + byte_off);
}
// end of a run of raw_issue calls: nothing is scheduled across this point, so the whole run is
// issued before the first use of any of it (N documents how many LATER loads/stores may still be
// in flight when this trip is consumed; the compiler derives the s_waitcnt itself)
template <int N>
__device__ __forceinline__ void raw_wait() {
  __builtin_amdgcn_sched_barrier(0);
}
template <int BYTES>
__device__ __forceinline__ void raw_land(RawReg<BYTES> &) {}
// W consecutive reals / int32 / int8 of a landed register image, as doubles / ints
template <int W>
__device__ __forceinline__ void raw_get(const RawReg<8 * W> &r, const double *, double (&o)[W]) {
  if constexpr (W == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 v = __builtin_bit_cast(d2, r.v);
    o[0] = v.x, o[1] = v.y;
  } else {
    static_assert(W == 1, "fp64: 1 or 2 rows per lane");
    o[0] = __builtin_bit_cast(double, r.v);
  }
}
template <int W>
__device__ __forceinline__ void raw_get(const RawReg<4 * W> &r, const float *, double (&o)[W]) {
  if constexpr (W == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_bit_cast(f4, r.v);
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
  } else if constexpr (W == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = __builtin_bit_cast(f2, r.v);
    o[0] = v.x, o[1] = v.y;
  } else {
    static_assert(W == 1, "fp32: 1, 2 or 4 rows per lane");
    o[0] = __builtin_bit_cast(float, r.v);
  }
}
template <int W>
__device__ __forceinline__ void raw_geti(const RawReg<4 * W> &r, const int32_t *, int (&o)[W]) {
  if constexpr (W == 4) {
    o[0] = r.v.x, o[1] = r.v.y, o[2] = r.v.z, o[3] = r.v.w;
  } else if constexpr (W == 2) {
    o[0] = r.v.x, o[1] = r.v.y;
  } else {
    o[0] = r.v;
  }
}
template <int W>
__device__ __forceinline__ void raw_geti(const RawReg<W> &r, const int8_t *, int (&o)[W]) {
#pragma unroll
  for (int k = 0; k < W; ++k) o[k] = (int)(signed char)((unsigned)r.v >> (8 * k));
}
// register image of W elements of type E
template <typename E, int W>
using RawOf = RawReg<(int)sizeof(E) * W>;

// ---- lane pairs (lane ^ 1): DPP quad_perm [1,0,3,2], no LDS traffic ----
// (bound_ctrl = 1 with full row / bank masks: every lane is written and no source lane is out of range, so the
//  result is the same -- but the compiler then knows the destination's old value is dead and does not
//  initialise it: one v_mov_b32 less per moved register)
__device__ __forceinline__ int pair_xchg(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ double pair_xchg(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)pair_xchg((int)b), hi = (unsigned)pair_xchg((int)(b >> 32));
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float pair_xchg(float v) {
  return __builtin_bit_cast(float, pair_xchg(__builtin_bit_cast(int, v)));
}
// The lanes l and l + 32 of a wave as a pair (gfx950: v_permlane32_swap): lo = the value of the pair's lane below
// 32, hi = the value of the one above, in BOTH lanes -- one instruction per 32 bits, no select behind it.
__device__ __forceinline__ void half_pair(int v, int &lo, int &hi) {
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  lo = (int)r[0], hi = (int)r[1];
}
__device__ __forceinline__ void half_pair(int64_t v, int64_t &lo, int64_t &hi) {
  int l0, h0, l1, h1;
  half_pair((int)v, l0, h0);
  half_pair((int)(v >> 32), l1, h1);
  lo = (int64_t)(((unsigned long long)(unsigned)l1 << 32) | (unsigned)l0);
  hi = (int64_t)(((unsigned long long)(unsigned)h1 << 32) | (unsigned)h0);
}
__device__ __forceinline__ void half_pair(double v, double &lo, double &hi) {
  int64_t l, h;
  half_pair(__builtin_bit_cast(int64_t, v), l, h);
  lo = __builtin_bit_cast(double, l), hi = __builtin_bit_cast(double, h);
}
__device__ __forceinline__ void half_pair(float v, float &lo, float &hi) {
  int l, h;
  half_pair(__builtin_bit_cast(int, v), l, h);
  lo = __builtin_bit_cast(float, l), hi = __builtin_bit_cast(float, h);
}
// rows per lane for kernels unrolled to MC column pairs: 16 B per lane per array, halved
// for MC >= 20 so that the 2*MC operand values of a row group still fit the register file
template <typename T, int MC>
struct RowsPer {
  static constexpr int V = MC >= 20 ? VecOf<T>::V / 2 : VecOf<T>::V;
};
// the same for kernels that also carry NACC fp64 accumulators per lane: rows per lane are halved
// while operands (2*MC*V values) + accumulators would not fit the 256 VGPRs (fp32: 4 rows of 20
// operands + 60 sums spill to AGPRs and run one wave per SIMD; 2 rows fit)
template <typename T, int MC, int NACC>
struct RowsPerAcc {
  static constexpr int V0 = RowsPer<T, MC>::V;
  // (never below 8 bytes per lane: 4-byte loads cost more than the spills they avoid)
  static constexpr int V =
      (V0 > 1 && (V0 / 2) * (int)sizeof(T) >= 8 && 2 * (2 * MC * V0 + NACC) > 240) ? V0 / 2 : V0;
};

// Grid-stride over rows in groups of V (16 B per lane per array by default), scalar tail.
// f(i, WTag<W>) handles rows i .. i+W-1.
template <typename T, int V = VecOf<T>::V, typename F>
__device__ __forceinline__ void for_rows(int64_t n, F &&f) {
  const int64_t nv = n / V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t iv = t0; iv < nv; iv += stride) f(iv * V, WTag<V>{});
  const int64_t it = nv * V + t0;
  if (it < n) f(it, WTag<1>{});
}

// Grid-stride over row groups for the kernels that schedule their own loads.  TripV / Trip1 hold
// the register images of one trip (V rows per lane / the scalar tail) and provide
//   static constexpr int NL        loads issued per trip
//   void issue(const Ctx &, int64_t i)   start every load of rows i .. i+W-1
//   void land()                          after the wait: hand the registers to the compiler
// f(trip, i, WTag<W>) consumes a landed trip; NS = the number of store instructions EVERY call of
// f issues (a lower bound is safe: the counted wait then merely waits for a few stores too).
// PIPE = false: issue, wait for everything, compute -- waves that share a SIMD overlap each other.
// PIPE = true : two trips in flight per wave -- the next trip's loads are issued BEFORE this
//   trip is computed, and the wait is a counted vmcnt that leaves them (and this trip's stores)
//   in flight.  For kernels that hold > 256 registers and therefore run ONE wave per SIMD, where
//   nothing else would cover the load latency (fp32 / fp64 at m = 20).  A trip past the end
//   re-reads the lane's last valid rows (no branch around the loads).
template <typename TripV, typename Trip1, int V, bool PIPE, int NS, typename Ctx, typename F>
__device__ __forceinline__ void for_rows_raw(int64_t n, const Ctx &c, F &&f) {
  const int64_t nv = n / V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (PIPE) {
    static_assert(TripV::NL + NS <= 63, "vmcnt counts to 63");
    if (t0 < nv) {
      TripV A, B;
      A.issue(c, t0 * V);
      int64_t n1 = t0 + stride;
      B.issue(c, (n1 < nv ? n1 : t0) * V);
      raw_wait<TripV::NL>();  // A has landed; B's loads stay in flight (no stores issued yet)
      A.land();
      f(A, t0 * V, WTag<V>{});
      while (n1 < nv) {  // B holds trip n1
        const int64_t n2 = n1 + stride;
        A.issue(c, (n2 < nv ? n2 : n1) * V);
        raw_wait<TripV::NL + NS>();  // older than A's loads: B's loads (now done), <= NS stores
        B.land();
        f(B, n1 * V, WTag<V>{});
        if (n2 >= nv) break;
        const int64_t n3 = n2 + stride;
        B.issue(c, (n3 < nv ? n3 : n2) * V);
        raw_wait<TripV::NL + NS>();
        A.land();
        f(A, n2 * V, WTag<V>{});
        n1 = n3;
      }
      raw_wait<0>();  // the last prefetch is unused: land it before its registers are reused
      A.land();
      B.land();
    }
  } else {
    for (int64_t iv = t0; iv < nv; iv += stride) {
      TripV A;
      A.issue(c, iv * V);
      raw_wait<0>();
      A.land();
      f(A, iv * V, WTag<V>{});
    }
  }
  const int64_t it = nv * V + t0;
  if (it < n) {
    Trip1 C;
    C.issue(c, it);
    raw_wait<0>();
    C.land();
    f(C, it, WTag<1>{});
  }
}

// ---- the passes over W under the tile-local free-row layout (WStore::lmask, "compact W") ----
// A wave takes one aligned tile of 128 rows per trip; lane l owns rows l and l + 64 of it (NOT 2l, 2l + 1: the rows
// whose layout bit is set sit at the front of the tile in ascending order, so the lanes that own such a row read ONE
// contiguous run of the column with each instruction -- measured against the lane-pair form, which reads every other
// element of the run twice: store pass 3.50 -> 2.80 ms, update pass 2.23 -> 1.95 ms at half of the rows free,
// n = 1e8; profiles/round6_a_compact_shapes_ab.txt).  The slots of the tile's rows follow from its two mask words,
// which are wave-uniform (scalar loads, fetched one issue ahead); a row's n-vector operands (x, g, ...) stay in
// natural order (8-byte loads, 512 contiguous bytes per wave instruction).
// A trip's two rows are handed to the kernel body as ONE row group of width 2 (the register images of a 16-byte
// load, filled by two 8-byte loads), so the body is the one the natural-order kernels run; what differs is where a
// row group's rows live (CwPairRows, kernels_common.hpp).  Rows that need their W entries although their layout bit is clear (the status
// changed since the layout was made) fetch them inside the body (Trip::reload_cols): the layout decides how many
// bytes move, never a result.  The last, partial tile (n not a multiple of 128) is taken row by row (Trip1).
// a full tile as its wave sees it: everything here is wave-uniform (scalar registers); a lane derives its two rows
// (tb + lane, tb + lane + 64), their in-tile slots and layout bits from it where it needs them
struct CwTile {
  int64_t tb;       // first row of the tile
  uint64_t m0, m1;  // layout bits of its rows [0, 64), [64, 128)
};
// in-tile slots (0 .. 127) and layout bits of the rows lane, lane + 64
__device__ __forceinline__ void cw_slots(const CwTile &t, int (&sl)[2], bool (&lf)[2]) {
  const int lane = (int)(threadIdx.x & 63);
  const int c0 = __popcll(t.m0), tf = c0 + __popcll(t.m1);
  // layout-free rows of the tile in front of this lane's two rows
  const int b0 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(t.m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)t.m0, 0u));
  const int b1 = c0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(t.m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)t.m1, 0u));
  lf[0] = (t.m0 >> lane) & 1ull, lf[1] = (t.m1 >> lane) & 1ull;
  sl[0] = lf[0] ? b0 : tf + (lane - b0);
  sl[1] = lf[1] ? b1 : tf + (64 + lane - b1);
}
__device__ __forceinline__ int64_t wave_uniform(int64_t v) {
  return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)v));
}
// Trip2: issue_cw(const Ctx &, const CwTile &), land();  f(trip, first row of the tile, WTag<2>)
// Trip1: issue_cw(const Ctx &, int64_t i, int64_t slot, bool lf, int64_t tile_first), land();  f(trip, i, WTag<1>)
template <typename Trip2, typename Trip1, typename Ctx, typename F>
__device__ __forceinline__ void for_tiles_cw(int64_t n, const Ctx &c, const uint64_t *__restrict__ lmask, F &&f) {
  const int lane = threadIdx.x & 63;
  const int64_t nfull = n >> 7;
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
  // (wave-uniform, which the compiler cannot see: scalar registers for the tile arithmetic and the mask words)
  const int64_t t0 = wave_uniform((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  // (one trip in flight per wave: two -- as for_rows_raw's PIPE -- lost against two or three waves per SIMD in every
  //  kernel of the layout, 3.2 vs 2.55 ms for the update pass at n = 1e8: HISTORY round 6)
  if (t0 < nfull) {
    uint64_t m0 = lmask[2 * t0], m1 = lmask[2 * t0 + 1];
    for (int64_t tr = t0;;) {
      const int64_t nx = tr + stride;
      const bool more = nx < nfull;
      const int64_t nxc = more ? nx : tr;
      const uint64_t m0n = lmask[2 * nxc], m1n = lmask[2 * nxc + 1];  // the next trip's words
      Trip2 A;
      A.issue_cw(c, CwTile{tr << 7, m0, m1});
      raw_wait<0>();
      A.land();
      f(A, tr << 7, WTag<2>{});
      if (!more) break;
      tr = nx, m0 = m0n, m1 = m1n;
    }
  }
  // the partial tile behind the full ones, row by row
  if ((n & 127) != 0 && t0 == nfull % stride) {
    const CwTile t{nfull << 7, lmask[2 * nfull], lmask[2 * nfull + 1]};
    int sl[2];
    bool lf[2];
    cw_slots(t, sl, lf);
    const int64_t i0 = t.tb + lane, i1 = i0 + 64;
    const bool v0 = i0 < n, v1 = i1 < n;  // (rows beyond n have no layout bit)
    Trip1 A, B;
    A.issue_cw(c, v0 ? i0 : n - 1, t.tb + sl[0], lf[0], t.tb);
    B.issue_cw(c, v1 ? i1 : n - 1, t.tb + sl[1], lf[1], t.tb);
    raw_wait<0>();
    A.land();
    B.land();
    if (v0) f(A, i0, WTag<1>{});
    if (v1) f(B, i1, WTag<1>{});
  }
}

// The same layout walked in HALF tiles: a wave takes 64 rows per trip, lane l owns row l of the half -- one row per
// lane and trip, as the natural-order kernels with many accumulators run (RowsPerAcc): the update pass with formk's
// new-row sums, whose lane pairs share the column accumulators (UpdScanPairTripCW) and so keep two waves per SIMD.
// Trip1 as above; f(trip, i, WTag<1>).  The partial half behind the full ones is taken by itself.
template <typename Trip1, typename Ctx, typename F>
__device__ __forceinline__ void for_halves_cw(int64_t n, const Ctx &c, const uint64_t *__restrict__ lmask, F &&f) {
  const int lane = threadIdx.x & 63;
  const int64_t nh = n >> 6;  // full halves
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t t0 = wave_uniform((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  // start every load of half h (its tile's mask words m0, m1 are wave-uniform)
  auto issue = [&](Trip1 &A, int64_t h, uint64_t m0, uint64_t m1, int64_t row_limit) {
    const bool hi = (h & 1) != 0;  // (uniform)
    const int c0 = __popcll(m0), tf = c0 + __popcll(m1);
    const uint64_t mw = hi ? m1 : m0;
    const int b = (hi ? c0 : 0) +
                  (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw, 0u));
    const bool lf = (mw >> lane) & 1ull;
    const int r = (hi ? 64 : 0) + lane;
    const int64_t tb = (h >> 1) << 7;
    int64_t i = (h << 6) + lane;
    if (i >= row_limit) i = row_limit - 1;
    A.issue_cw(c, i, tb + (lf ? b : tf + (r - b)), lf, tb);
  };
  if (t0 < nh) {
    const int64_t last = nh - 1;
    auto cl = [&](int64_t h) { return h < nh ? h : last; };
    uint64_t m0 = lmask[(t0 >> 1) * 2], m1 = lmask[(t0 >> 1) * 2 + 1];
    for (int64_t h = t0;;) {
      const int64_t nx = h + stride, nxc = cl(nx);
      const uint64_t m0n = lmask[(nxc >> 1) * 2], m1n = lmask[(nxc >> 1) * 2 + 1];  // the next trip's words
      Trip1 A;
      issue(A, h, m0, m1, n);
      raw_wait<0>();
      A.land();
      f(A, (h << 6) + lane, WTag<1>{});
      if (nx >= nh) break;
      h = nx, m0 = m0n, m1 = m1n;
    }
  }
  if ((n & 63) != 0 && t0 == nh % stride) {  // the partial half
    const uint64_t m0 = lmask[(nh >> 1) * 2], m1 = lmask[(nh >> 1) * 2 + 1];
    Trip1 A;
    issue(A, nh, m0, m1, n);
    raw_wait<0>();
    A.land();
    if ((nh << 6) + lane < n) f(A, (nh << 6) + lane, WTag<1>{});
  }
}

// ---- reductions: DPP inside the 16-lane rows of a wave, LDS across rows and waves ----
// Round 3 reduced every slot with six dependent __shfl_down steps: each step is two ds_bpermute_b32,
// an s_waitcnt lgkmcnt(0) and an add, and the compiler kept the slots' chains one behind the other --
// 570 serialised LDS round trips for the 95 sums of the update pass, 25-30 us per wave.  At n = 1e8
// that is 3 % of the pass; at the 8-GPU per-rank shape (1.25e7 rows) it was 70 of its 410 us, at
// n = 1e6 70 of 100.  Now nothing in the epilogue waits for LDS:
//   * inside a 16-lane row values move by DPP (quad_perm, row_half_mirror, row_mirror: plain VALU
//     moves, no LDS);
//   * many sums (K >= 16) are reduce-SCATTERed: at each of the four steps a lane hands one half of
//     its slots to its partner and keeps the other, so the work halves every step (2 K adds per
//     lane in all instead of 6 K) and lane j of a row ends up with K/16 slots summed over the row;
//   * the 4 rows x 4 waves = 16 row results per slot go through LDS once and are added in a fixed
//     order by one thread per slot.
// Fixed shape => deterministic for a fixed (n, grid), as before; the ORDER of the additions inside
// a workgroup differs from round 3's (reductions are compared at 1e-10, never bit for bit against
// the oracle).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {  // every lane: the value of its DPP source lane
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7 - i inside each group of 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15 - i inside each row of 16
constexpr int DPP_ROR8 = 0x128;        // row_ror:8: lane i <-> i xor 8 inside each row of 16
// OP: 0 sum, 1 min, 2 max
template <int OP>
__device__ __forceinline__ double red_op(double a, double b) {
  return OP == 0 ? a + b : (OP == 1 ? fmin(a, b) : fmax(a, b));
}
// every lane of a 16-lane row receives the row's result (a + b == b + a bit for bit, so all 16 agree)
template <int OP>
__device__ __forceinline__ double row_reduce(double v) {
  v = red_op<OP>(v, dpp_mov<DPP_XOR1>(v));
  v = red_op<OP>(v, dpp_mov<DPP_XOR2>(v));
  v = red_op<OP>(v, dpp_mov<DPP_HALF_MIRROR>(v));
  v = red_op<OP>(v, dpp_mov<DPP_MIRROR>(v));
  return v;
}
__device__ __forceinline__ double lane_bcast(double v, int lane) {  // wave-uniform copy of one lane's value
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
// whole-wave results, valid in EVERY lane (rows added in the order 0, 1, 2, 3)
template <int OP>
__device__ __forceinline__ double wave_reduce(double v) {
  v = row_reduce<OP>(v);
  return red_op<OP>(red_op<OP>(red_op<OP>(lane_bcast(v, 0), lane_bcast(v, 16)), lane_bcast(v, 32)),
                    lane_bcast(v, 48));
}
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<0>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<1>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<2>(v); }

// one reduce-scatter step: N slots in, N / 2 out.  `up` = this lane's side of the exchange (the bit
// of its lane number that the DPP pattern CTRL flips); a lane keeps in[p] (down side) or
// in[p + N/2] (up side) and receives the same slot from its partner.
template <int N, int CTRL>
__device__ __forceinline__ void row_halve(const double (&in)[N], double (&out)[N / 2], bool up) {
  static_assert(N % 2 == 0, "even slot count");
#pragma unroll
  for (int p = 0; p < N / 2; ++p) {
    const double keep = up ? in[p + N / 2] : in[p];
    const double send = up ? in[p] : in[p + N / 2];
    out[p] = keep + dpp_mov<CTRL>(send);
  }
}

// acc[0..nsum) are sums, then nmin minima, then nmax maxima (constants at every call site: the
// conditions below fold away).  One value per slot per workgroup goes to
// part[slot*pstride + blockIdx.x].  Every lane of the workgroup must call this.
template <int K>
__device__ __forceinline__ void block_reduce_store(const double (&acc)[K], int nsum, int nmin,
                                                   int nmax, double *__restrict__ part,
                                                   int pstride) {
  constexpr bool SCATTER = K >= 16;
  constexpr int KP = SCATTER ? (K + 15) / 16 * 16 : K;
  __shared__ double sm[16][KP];  // [wave * 4 + row][slot]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rr = w * 4 + (lane >> 4);
  const int ntot = nsum + nmin + nmax;
  if constexpr (SCATTER) {
    double t0[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) t0[k] = k < K ? acc[k] : 0.0;
    double t1[KP / 2], t2[KP / 4], t3[KP / 8], t4[KP / 16];
    // Partners must hold the SAME slots, i.e. differ in exactly the bit the step splits on.  The
    // mirror patterns flip every lower bit too, so the half-mirror step (bit 2, partner 7 - i) comes
    // FIRST, while every lane still holds all slots; bits 0 and 1 are exact xor exchanges (quad_perm),
    // bit 3 is a rotation by 8 inside the row (i +- 8 mod 16 = i xor 8).
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    row_halve<KP, DPP_HALF_MIRROR>(t0, t1, b2);
    row_halve<KP / 2, DPP_XOR1>(t1, t2, b0);
    row_halve<KP / 4, DPP_XOR2>(t2, t3, b1);
    row_halve<KP / 8, DPP_ROR8>(t3, t4, b3);
    // this lane's slots: p + (KP/2) b2 + (KP/4) b0 + (KP/8) b1 + (KP/16) b3
    const int base = (b2 ? KP / 2 : 0) + (b0 ? KP / 4 : 0) + (b1 ? KP / 8 : 0) + (b3 ? KP / 16 : 0);
#pragma unroll
    for (int p = 0; p < KP / 16; ++p)
      if (base + p < nsum) sm[rr][base + p] = t4[p];
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k < nsum) {
        const double v = row_reduce<0>(acc[k]);
        if ((lane & 15) == 0) sm[rr][k] = v;
      }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k >= nsum && k < ntot) {
      const double v = k < nsum + nmin ? row_reduce<1>(acc[k]) : row_reduce<2>(acc[k]);
      if ((lane & 15) == 0) sm[rr][k] = v;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ntot; k += blockDim.x) {
    double s = sm[0][k];
    if (k < nsum) {
#pragma unroll
      for (int q = 1; q < 16; ++q) s = s + sm[q][k];
    } else if (k < nsum + nmin) {
#pragma unroll
      for (int q = 1; q < 16; ++q) s = fmin(s, sm[q][k]);
    } else {
#pragma unroll
      for (int q = 1; q < 16; ++q) s = fmax(s, sm[q][k]);
    }
    part[(size_t)k * pstride + blockIdx.x] = s;
  }
}

}  // namespace lbk
