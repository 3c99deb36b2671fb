// Bare-kernel A/B for TILE-LOCAL FREE-ROW COMPACTION of W (companion of pass_shapes.hip).
//
// The reference's cmprlb / subsm / formk loops run over the FREE rows only (src/lbfgsb.f90:1565-1583, :2743-2778,
// :1756-1793 through Index, :2044-2054); the library's two passes stream all n rows of every W column under a
// mask.  Candidate layout: inside fixed tiles of T rows every W column stores the tile's layout-free rows first
// (ascending = Index order), the other rows behind them.  x, g, t, r, iwhere keep their natural order.  A row's
// slot is   tile * T + (free ? free_before : tile_free + (row_in_tile - free_before))   with
// free_before = (layout-free rows of the tile before this row) = ginfo[group].x + popcount of the wave's 128-bit
// layout mask below the row: 16 + 8 bytes of SCALAR loads per 128 rows.  Correct for ANY iwhere (a row that is
// needed but sits in the tail is fetched from the tail): the layout only decides how many bytes move.
//
// Shapes timed (n rows, fp64, 9 stored pairs + the pending pair, as the headline's steady state):
//   store pass (subsm_update_kernel):  x, g, r, t, iwhere | 18 W columns on needed rows | stores x', Wy-col, Ws-col
//   update pass (update_scan_kernel):  x, g, r, t, iwhere | 18 W columns on needed rows | 8 sums per column
// in three forms each:
//   masked    today's: one lane = rows (2l, 2l+1), 16-byte loads of every column for every row
//   c_pair    compact, lane = rows (2l, 2l+1): 16-byte loads of the vectors, two 8-byte gathers per column
//   c_split   compact, lane = rows (l, l + 64) of the wave's 128: 8-byte loads throughout; a wave instruction
//             reads ONE contiguous run of the free region
// at free fractions 1.0 / 0.5 / 0.1 (pseudo-random pattern, like the headline's 49 999 496 of 1e8) and tiles of
// 128 / 1024 / 4096 rows; plus "stale": 0.5 % of the rows changed status since the layout was made.
// Every form's checksum is compared with the masked one (same sums in another order: 1e-9).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 compact_shapes.hip -o bin/compact_shapes
//   bin/compact_shapes [n = 100000000] [reps = 5]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int NC = 9;       // stored pairs read from W (the 10th is pending in r, t)
constexpr int BLOCK = 256;

__device__ __forceinline__ d2 ldnt2(const double *p) { return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p)); }
__device__ __forceinline__ double ldnt1(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt2(double *p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p)); }
__device__ __forceinline__ void stnt1(double *p, double v) { __builtin_nontemporal_store(v, p); }

__host__ __device__ inline uint32_t hash32(uint64_t k) {
  k ^= k >> 33, k *= 0xff51afd7ed558ccdull, k ^= k >> 33, k *= 0xc4ceb9fe1a85ec53ull, k ^= k >> 33;
  return (uint32_t)k;
}
__host__ __device__ inline double wval(int c, int64_t row) { return (double)(hash32((uint64_t)row * 64 + c) & 0xffff) * (1.0 / 65536.0) - 0.5; }

struct Layout {
  const uint64_t *lmask;  // 1 bit per row: layout-free
  const uint2 *ginfo;     // per 128-row group: x = layout-free rows of its tile before the group, y = the tile's count
  int tshift;             // log2(T)
};


// trip -> wave assignment: round k of the grid-stride loop is rotated by k * g_rot waves (0: the plain grid-stride
// loop, every wave marching through memory in lock-step with the same stride)
__device__ int g_rot;
#define TRIPS(tr, ntr)                                                                                              \
  for (int64_t k_ = 0, nw_ = (int64_t)gridDim.x * 4, wid_ = (int64_t)blockIdx.x * 4 + wv, tr = wid_; k_ * nw_ < (ntr); \
       ++k_, tr = k_ * nw_ + (wid_ + k_ * g_rot) % nw_)                                                              \
    if (tr < (ntr))
#define LANES(iv, nv)                                                                                                 \
  for (int64_t k_ = 0, nt_ = (int64_t)gridDim.x * BLOCK, tid_ = (int64_t)blockIdx.x * BLOCK + threadIdx.x, iv = tid_;  \
       k_ * nt_ < (nv); ++k_, iv = k_ * nt_ + (tid_ + k_ * g_rot * 64) % nt_)                                          \
    if (iv < (nv))
// ---- set-up ----
__global__ void k_fill_vec(int64_t n, double *x, double *g, double *r, double *t) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] = wval(40, i), g[i] = wval(41, i), r[i] = wval(42, i), t[i] = wval(43, i);
  }
}
__global__ void k_mask(int64_t n, uint32_t thresh, uint32_t stale_thresh, uint64_t *lmask, int8_t *iw) {
  // one thread per 64 rows
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 64 >= n) return;
  uint64_t m = 0;
  for (int b = 0; b < 64; ++b) {
    const int64_t i = w * 64 + b;
    if (i >= n) break;
    const bool fr = hash32((uint64_t)i * 2654435761ull + 17) <= thresh;
    if (fr) m |= 1ull << b;
    const bool flip = stale_thresh && hash32((uint64_t)i * 40503ull + 99) < stale_thresh;
    iw[i] = (fr != flip) ? 0 : 1;   // iwhere <= 0: free NOW
  }
  lmask[w] = m;
}
__global__ void k_ginfo(int64_t ngroups, int gpt, const uint64_t *lmask, uint2 *ginfo) {
  // one thread per tile
  const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t g0 = tile * gpt;
  if (g0 >= ngroups) return;
  uint32_t run = 0;
  for (int k = 0; k < gpt && g0 + k < ngroups; ++k) {
    ginfo[g0 + k].x = run;
    run += __popcll(lmask[2 * (g0 + k)]) + __popcll(lmask[2 * (g0 + k) + 1]);
  }
  for (int k = 0; k < gpt && g0 + k < ngroups; ++k) ginfo[g0 + k].y = run;
}
__device__ __forceinline__ int64_t slot_of(int64_t row, const Layout &L) {
  const int64_t grp = row >> 7;
  const int r = (int)(row & 127);
  const uint64_t m0 = L.lmask[2 * grp], m1 = L.lmask[2 * grp + 1];
  const uint2 gi = L.ginfo[grp];
  int before;
  bool fr;
  if (r < 64) before = __popcll(m0 & ((1ull << r) - 1ull)), fr = (m0 >> r) & 1;
  else before = __popcll(m0) + __popcll(m1 & ((1ull << (r - 64)) - 1ull)), fr = (m1 >> (r - 64)) & 1;
  const int64_t T = 1ll << L.tshift, tile = row >> L.tshift, rit = row & (T - 1);
  const int64_t fb = gi.x + before;
  return tile * T + (fr ? fb : (int64_t)gi.y + (rit - fb));
}
__global__ void k_fill_w(int64_t n, int64_t ld, double *wn, double *wc, Layout L) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = wc ? slot_of(i, L) : 0;
    for (int c = 0; c < 2 * NC; ++c) {
      const double v = wval(c, i);
      if (wn) wn[(int64_t)c * ld + i] = v;
      if (wc) wc[(int64_t)c * ld + s] = v;
    }
  }
}

// ---- per-row arithmetic (the same in every form) ----
struct Coefs { double c[2 * NC], w[2 * NC]; };
__device__ __forceinline__ double newton_row(double x, double g, const double (&a)[NC], const double (&b)[NC], const Coefs &cf) {
  double dk = -0.7 * (x * 0.25) - g;
#pragma unroll
  for (int j = 0; j < NC; ++j) dk = dk + a[j] * cf.c[j] + b[j] * cf.c[NC + j];
#pragma unroll
  for (int j = 0; j < NC; ++j) dk = dk + a[j] * cf.w[j] * 1.25 + b[j] * cf.w[NC + j];
  return 0.8 * dk;
}

__device__ __forceinline__ void wave_sum_store(double v, double *out) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, v);
}

// ============================ store pass ============================
// masked: today's shape
__global__ __launch_bounds__(BLOCK) void store_masked(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, int64_t ld, Coefs cf, double *xout, double *cwy, double *cws, double *sums) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * BLOCK;
  double acc = 0.0;
  LANES(iv, nv) {
    const int64_t i = iv * 2;
    const d2 xv = ldnt2(x + i), gv = ldnt2(g + i), rv = ldnt2(r + i), tv = ldnt2(t + i);
    const char2 iv2 = *reinterpret_cast<const char2 *>(iw + i);
    d2 a[NC], b[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) a[j] = ldnt2(w + (int64_t)j * ld + i), b[j] = ldnt2(w + (int64_t)(NC + j) * ld + i);
    __builtin_amdgcn_sched_barrier(0);
    d2 z = xv;
    const int fw[2] = {iv2.x, iv2.y};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (fw[k] <= 0) {
        double ak[NC], bk[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
        z[k] = fmin(1.0, fmax(-1.0, xv[k] + newton_row(xv[k], gv[k], ak, bk, cf)));
      }
      const double dv = z[k] - xv[k];
      acc = acc + dv * gv[k];
    }
    stnt2(xout + i, z);
    stnt2(cwy + i, gv - rv);
    stnt2(cws + i, xv - tv);
  }
  wave_sum_store(acc, sums);
}

// one wave-trip = 128 consecutive rows; its layout words are wave-uniform (scalar loads)
struct WaveTile {
  uint64_t m0, m1;
  uint32_t gb, tf;
  int64_t tbase;   // tile * T
  int64_t git;     // first row of the group inside its tile
};
__device__ __forceinline__ WaveTile wave_tile(int64_t row0, const Layout &L) {
  const int64_t grp = __builtin_amdgcn_readfirstlane((int)(row0 >> 7));   // (n < 2^38 rows: the group index fits 31 bits)
  WaveTile wt;
  wt.m0 = L.lmask[2 * grp], wt.m1 = L.lmask[2 * grp + 1];
  const uint2 gi = L.ginfo[grp];
  wt.gb = gi.x, wt.tf = gi.y;
  const int64_t T = 1ll << L.tshift;
  wt.tbase = ((grp << 7) >> L.tshift) << L.tshift;
  wt.git = (grp << 7) & (T - 1);
  return wt;
}

// compact, lane = rows (2l, 2l + 1)
__global__ __launch_bounds__(BLOCK) void store_cpair(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, Coefs cf, double *xout, double *py, double *ps, double *sums) {
  const int64_t ntr = n / 128;  // (n a multiple of 128 here)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc = 0.0;
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 128, i = row0 + 2 * lane;
    const WaveTile wt = wave_tile(row0, L);
    const d2 xv = ldnt2(x + i), gv = ldnt2(g + i), rv = ldnt2(r + i), tv = ldnt2(t + i);
    const char2 iv2 = *reinterpret_cast<const char2 *>(iw + i);
    const uint64_t mm = lane < 32 ? wt.m0 : wt.m1;
    const int sh = (2 * lane) & 63;
    const int before = (lane < 32 ? 0 : __popcll(wt.m0)) + __popcll(mm & ((1ull << sh) - 1ull));
    const int f0 = (int)((mm >> sh) & 1), f1 = (int)((mm >> (sh + 1)) & 1);
    const int64_t fb0 = wt.gb + before, fb1 = fb0 + f0;
    const int64_t rit0 = wt.git + 2 * lane;
    const int64_t s0 = wt.tbase + (f0 ? fb0 : (int64_t)wt.tf + (rit0 - fb0));
    const int64_t s1 = wt.tbase + (f1 ? fb1 : (int64_t)wt.tf + (rit0 + 1 - fb1));
    const bool need0 = iv2.x <= 0, need1 = iv2.y <= 0;
    double a[NC][2], b[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *pa = w + (int64_t)j * ld, *pb = w + (int64_t)(NC + j) * ld;
      a[j][0] = ldnt1(need0 ? pa + s0 : zero), a[j][1] = ldnt1(need1 ? pa + s1 : zero);
      b[j][0] = ldnt1(need0 ? pb + s0 : zero), b[j][1] = ldnt1(need1 ? pb + s1 : zero);
    }
    __builtin_amdgcn_sched_barrier(0);
    d2 z = xv;
    const bool need[2] = {need0, need1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (need[k]) {
        double ak[NC], bk[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
        z[k] = fmin(1.0, fmax(-1.0, xv[k] + newton_row(xv[k], gv[k], ak, bk, cf)));
      }
      const double dv = z[k] - xv[k];
      acc = acc + dv * gv[k];
    }
    stnt2(xout + i, z);
    const d2 yn = gv - rv, sn = xv - tv;
    stnt1(py + s0, yn[0]), stnt1(py + s1, yn[1]);
    stnt1(ps + s0, sn[0]), stnt1(ps + s1, sn[1]);
  }
  wave_sum_store(acc, sums);
}

// compact, lane = rows (l, l + 64) of the wave's 128
// LIB: as the library's pass does it -- the W loads go by the layout bits alone (no load waits for iwhere; a row
// without a bit reads the first entry of its group's run) and are plain, not nontemporal
template <bool LIB>
__global__ __launch_bounds__(BLOCK) void store_csplit(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, Coefs cf, double *xout, double *py, double *ps, double *sums) {
  const int64_t ntr = n / 128;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc = 0.0;
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 128;
    const WaveTile wt = wave_tile(row0, L);
    const int64_t i0 = row0 + lane, i1 = i0 + 64;
    const double xv[2] = {ldnt1(x + i0), ldnt1(x + i1)}, gv[2] = {ldnt1(g + i0), ldnt1(g + i1)};
    const double rv[2] = {ldnt1(r + i0), ldnt1(r + i1)}, tv[2] = {ldnt1(t + i0), ldnt1(t + i1)};
    const int iw0 = iw[i0], iw1 = iw[i1];
    const uint64_t below = (1ull << lane) - 1ull;
    const int b0 = __popcll(wt.m0 & below), b1 = __popcll(wt.m0) + __popcll(wt.m1 & below);
    const int f0 = (int)((wt.m0 >> lane) & 1), f1 = (int)((wt.m1 >> lane) & 1);
    const int64_t fb0 = wt.gb + b0, fb1 = wt.gb + b1;
    const int64_t s0 = wt.tbase + (f0 ? fb0 : (int64_t)wt.tf + (wt.git + lane - fb0));
    const int64_t s1 = wt.tbase + (f1 ? fb1 : (int64_t)wt.tf + (wt.git + 64 + lane - fb1));
    const bool need[2] = {iw0 <= 0, iw1 <= 0};
    double a[NC][2], b[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *pa = w + (int64_t)j * ld, *pb = w + (int64_t)(NC + j) * ld;
      if constexpr (LIB) {
        const int64_t d0 = f0 ? s0 : wt.tbase + wt.gb, d1 = f1 ? s1 : wt.tbase + wt.gb;
        a[j][0] = pa[d0], a[j][1] = pa[d1], b[j][0] = pb[d0], b[j][1] = pb[d1];
      } else {
        a[j][0] = ldnt1(need[0] ? pa + s0 : zero), a[j][1] = ldnt1(need[1] ? pa + s1 : zero);
        b[j][0] = ldnt1(need[0] ? pb + s0 : zero), b[j][1] = ldnt1(need[1] ? pb + s1 : zero);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    double z[2] = {xv[0], xv[1]};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (need[k]) {
        double ak[NC], bk[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
        z[k] = fmin(1.0, fmax(-1.0, xv[k] + newton_row(xv[k], gv[k], ak, bk, cf)));
      }
      const double dv = z[k] - xv[k];
      acc = acc + dv * gv[k];
    }
    stnt1(xout + i0, z[0]), stnt1(xout + i1, z[1]);
    stnt1(py + s0, gv[0] - rv[0]), stnt1(py + s1, gv[1] - rv[1]);
    stnt1(ps + s0, xv[0] - tv[0]), stnt1(ps + s1, xv[1] - tv[1]);
  }
  wave_sum_store(acc, sums);
}


// ---- storing pass, compact, SLOT order (T = 128 only): see upd_dense ----
template <bool PLAIN>
__global__ __launch_bounds__(BLOCK) void store_dense(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, Coefs cf, double *xout, double *py, double *ps, double *sums) {
  __shared__ uint8_t perm[BLOCK / 64][128];
  const int64_t ntr = n / 128;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc = 0.0;
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 128;
    const WaveTile wt = wave_tile(row0, L);
    const uint64_t below = (1ull << lane) - 1ull;
    const int c0 = __popcll(wt.m0), tf = c0 + __popcll(wt.m1);
    const int b0 = __popcll(wt.m0 & below), b1 = c0 + __popcll(wt.m1 & below);
    const int s0 = ((wt.m0 >> lane) & 1) ? b0 : tf + (lane - b0);
    const int s1 = ((wt.m1 >> lane) & 1) ? b1 : tf + (64 + lane - b1);
    perm[wv][s0] = (uint8_t)lane, perm[wv][s1] = (uint8_t)(64 + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t i0 = row0 + perm[wv][lane], i1 = row0 + perm[wv][64 + lane];
    __builtin_amdgcn_wave_barrier();
    const double xv[2] = {ldnt1(x + i0), ldnt1(x + i1)}, gv[2] = {ldnt1(g + i0), ldnt1(g + i1)};
    const double rv[2] = {ldnt1(r + i0), ldnt1(r + i1)}, tv[2] = {ldnt1(t + i0), ldnt1(t + i1)};
    const bool need[2] = {lane < tf, 64 + lane < tf};   // (the layout was made from this iwhere: bit = free)
    double a[NC][2], b[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *pa = w + (int64_t)j * ld + row0, *pb = w + (int64_t)(NC + j) * ld + row0;
      a[j][0] = PLAIN ? *(need[0] ? pa + lane : zero) : ldnt1(need[0] ? pa + lane : zero), b[j][0] = PLAIN ? *(need[0] ? pb + lane : zero) : ldnt1(need[0] ? pb + lane : zero);
      a[j][1] = 0.0, b[j][1] = 0.0;
    }
    if (tf > 64) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const double *pa = w + (int64_t)j * ld + row0, *pb = w + (int64_t)(NC + j) * ld + row0;
        a[j][1] = *(need[1] ? pa + 64 + lane : zero), b[j][1] = *(need[1] ? pb + 64 + lane : zero);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    double z[2] = {xv[0], xv[1]};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (need[k]) {
        double ak[NC], bk[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
        z[k] = fmin(1.0, fmax(-1.0, xv[k] + newton_row(xv[k], gv[k], ak, bk, cf)));
      }
      const double dv = z[k] - xv[k];
      acc = acc + dv * gv[k];
    }
    stnt1(xout + i0, z[0]), stnt1(xout + i1, z[1]);
    stnt1(py + row0 + lane, gv[0] - rv[0]), stnt1(py + row0 + 64 + lane, gv[1] - rv[1]);
    stnt1(ps + row0 + lane, xv[0] - tv[0]), stnt1(ps + row0 + 64 + lane, xv[1] - tv[1]);
  }
  wave_sum_store(acc, sums);
}
// ============================ update pass ============================
// 8 sums per column (matupd 2, cauchy p 2, formk's new row 4) + a few row sums
struct UAcc {
  double s[8][NC];
  double misc[3];
};
__device__ __forceinline__ void uacc_zero(UAcc &A) {
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int j = 0; j < NC; ++j) A.s[q][j] = 0.0;
  A.misc[0] = A.misc[1] = A.misc[2] = 0.0;
}
__device__ __forceinline__ void urow(UAcc &A, double x, double g, double r, double t, int iw, const double (&a)[NC], const double (&b)[NC]) {
  const double s = x - t, y = g - r;
  const bool fr = iw <= 0;
  const double ng = fr ? -g : 0.0, yf = fr ? y : 0.0, sa = fr ? 0.0 : s;
  A.misc[0] += g * s, A.misc[1] += y * y, A.misc[2] -= ng * ng;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    A.s[0][j] += s * a[j], A.s[1][j] += b[j] * s, A.s[2][j] += a[j] * ng, A.s[3][j] += b[j] * ng;
    A.s[4][j] = __builtin_fma(yf, a[j], A.s[4][j]), A.s[5][j] = __builtin_fma(sa, b[j], A.s[5][j]);
    A.s[6][j] = __builtin_fma(sa, a[j], A.s[6][j]), A.s[7][j] = __builtin_fma(b[j], yf, A.s[7][j]);
  }
}
__device__ __forceinline__ void uacc_out(const UAcc &A, double *sums) {
  double tot = A.misc[0] + A.misc[1] + A.misc[2];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int j = 0; j < NC; ++j) tot += A.s[q][j] * (1.0 + 0.01 * (q * NC + j));
  wave_sum_store(tot, sums);
}

__global__ __launch_bounds__(BLOCK) void upd_masked(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, int64_t ld, double *sums) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * BLOCK;
  UAcc A;
  uacc_zero(A);
  LANES(iv, nv) {
    const int64_t i = iv * 2;
    const d2 xv = ldnt2(x + i), gv = ldnt2(g + i), rv = ldnt2(r + i), tv = ldnt2(t + i);
    const char2 iv2 = *reinterpret_cast<const char2 *>(iw + i);
    d2 a[NC], b[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) a[j] = ldnt2(w + (int64_t)j * ld + i), b[j] = ldnt2(w + (int64_t)(NC + j) * ld + i);
    __builtin_amdgcn_sched_barrier(0);
    const int fw[2] = {iv2.x, iv2.y};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double ak[NC], bk[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
      urow(A, xv[k], gv[k], rv[k], tv[k], fw[k], ak, bk);
    }
  }
  uacc_out(A, sums);
}

// needed rows of the update pass: s != 0 or the row is free at the new point (see the header of this file)
template <bool SPLIT>
__global__ __launch_bounds__(BLOCK) void upd_compact(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, double *sums) {
  const int64_t ntr = n / 128;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  UAcc A;
  uacc_zero(A);
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 128;
    const WaveTile wt = wave_tile(row0, L);
    double xv[2], gv[2], rv[2], tv[2];
    int fw[2];
    int64_t s0, s1;
    if constexpr (SPLIT) {
      const int64_t i0 = row0 + lane, i1 = i0 + 64;
      xv[0] = ldnt1(x + i0), xv[1] = ldnt1(x + i1), gv[0] = ldnt1(g + i0), gv[1] = ldnt1(g + i1);
      rv[0] = ldnt1(r + i0), rv[1] = ldnt1(r + i1), tv[0] = ldnt1(t + i0), tv[1] = ldnt1(t + i1);
      fw[0] = iw[i0], fw[1] = iw[i1];
      const uint64_t below = (1ull << lane) - 1ull;
      const int b0 = __popcll(wt.m0 & below), b1 = __popcll(wt.m0) + __popcll(wt.m1 & below);
      const int f0 = (int)((wt.m0 >> lane) & 1), f1 = (int)((wt.m1 >> lane) & 1);
      const int64_t fb0 = wt.gb + b0, fb1 = wt.gb + b1;
      s0 = wt.tbase + (f0 ? fb0 : (int64_t)wt.tf + (wt.git + lane - fb0));
      s1 = wt.tbase + (f1 ? fb1 : (int64_t)wt.tf + (wt.git + 64 + lane - fb1));
    } else {
      const int64_t i = row0 + 2 * lane;
      const d2 x2 = ldnt2(x + i), g2 = ldnt2(g + i), r2 = ldnt2(r + i), t2 = ldnt2(t + i);
      const char2 iv2 = *reinterpret_cast<const char2 *>(iw + i);
      xv[0] = x2.x, xv[1] = x2.y, gv[0] = g2.x, gv[1] = g2.y, rv[0] = r2.x, rv[1] = r2.y, tv[0] = t2.x, tv[1] = t2.y;
      fw[0] = iv2.x, fw[1] = iv2.y;
      const uint64_t mm = lane < 32 ? wt.m0 : wt.m1;
      const int sh = (2 * lane) & 63;
      const int before = (lane < 32 ? 0 : __popcll(wt.m0)) + __popcll(mm & ((1ull << sh) - 1ull));
      const int f0 = (int)((mm >> sh) & 1), f1 = (int)((mm >> (sh + 1)) & 1);
      const int64_t fb0 = wt.gb + before, fb1 = fb0 + f0;
      const int64_t rit0 = wt.git + 2 * lane;
      s0 = wt.tbase + (f0 ? fb0 : (int64_t)wt.tf + (rit0 - fb0));
      s1 = wt.tbase + (f1 ? fb1 : (int64_t)wt.tf + (rit0 + 1 - fb1));
    }
    // (a row needs its W entries if it moved (s != 0) or is free at the new point)
    const bool need0 = fw[0] <= 0 || xv[0] != tv[0], need1 = fw[1] <= 0 || xv[1] != tv[1];
    double a[NC][2], b[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *pa = w + (int64_t)j * ld, *pb = w + (int64_t)(NC + j) * ld;
      a[j][0] = ldnt1(need0 ? pa + s0 : zero), a[j][1] = ldnt1(need1 ? pa + s1 : zero);
      b[j][0] = ldnt1(need0 ? pb + s0 : zero), b[j][1] = ldnt1(need1 ? pb + s1 : zero);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double ak[NC], bk[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
      urow(A, xv[k], gv[k], rv[k], tv[k], fw[k], ak, bk);
    }
  }
  uacc_out(A, sums);
}

// ---- update pass, compact, lane PAIRS share the column sums over whole-tile trips (rows l, l + 64 per lane) ----
// Each lane of a pair (2p, 2p + 1) owns HALF of the columns (even lane: [0, 5), odd lane: [5, 9)) and sums them over
// the pair's FOUR rows; it loads its columns at the slots of all four rows (the partner's slots come over by
// shuffle) and receives the partner's row scalars.  8 x 5 instead of 8 x 9 column accumulators per lane.
template <int X>
__global__ __launch_bounds__(BLOCK) void upd_pair2(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, double *sums) {
  constexpr int H = 5;
  const int64_t ntr = n / 128;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool hi = lane & X;
  double acc[8][H];
  double misc[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int j = 0; j < H; ++j) acc[q][j] = 0.0;
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 128;
    const WaveTile wt = wave_tile(row0, L);
    const int64_t i0 = row0 + lane, i1 = i0 + 64;
    const double xv[2] = {ldnt1(x + i0), ldnt1(x + i1)}, gv[2] = {ldnt1(g + i0), ldnt1(g + i1)};
    const double rv[2] = {ldnt1(r + i0), ldnt1(r + i1)}, tv[2] = {ldnt1(t + i0), ldnt1(t + i1)};
    const int fw[2] = {iw[i0], iw[i1]};
    const uint64_t below = (1ull << lane) - 1ull;
    const int b0 = __popcll(wt.m0 & below), b1 = __popcll(wt.m0) + __popcll(wt.m1 & below);
    const int f0 = (int)((wt.m0 >> lane) & 1), f1 = (int)((wt.m1 >> lane) & 1);
    const int fb0 = wt.gb + b0, fb1 = wt.gb + b1;
    // slots relative to the tile base (T <= 4096: an int)
    // (from the layout bits alone, as the library's kernels do: no load depends on iwhere; a row without a bit reads
    //  the first entry of its group's run and multiplies it by exact zeros)
    const int own0 = f0 ? fb0 : (int)wt.gb, own1 = f1 ? fb1 : (int)wt.gb;
    const int oth0 = __shfl_xor(own0, X), oth1 = __shfl_xor(own1, X);
    // order of the pair's four rows: (even lane row0, odd lane row0, even lane row1, odd lane row1)
    const int sl[4] = {hi ? oth0 : own0, hi ? own0 : oth0, hi ? oth1 : own1, hi ? own1 : oth1};
    double a[H][4], b[H][4];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const int c = (hi ? H : 0) + j;
      const bool dead = c >= NC;
      const double *pa = w + (int64_t)(dead ? 0 : c) * ld + wt.tbase, *pb = w + (int64_t)(NC + (dead ? 0 : c)) * ld + wt.tbase;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool z = dead;
        a[j][k] = *(z ? zero : pa + sl[k]), b[j][k] = *(z ? zero : pb + sl[k]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    double s4[4], ng4[4], yf4[4], sa4[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double s = xv[k] - tv[k], y = gv[k] - rv[k];
      const bool fr = fw[k] <= 0;
      const double ng = fr ? -gv[k] : 0.0, yf = fr ? y : 0.0, sa = fr ? 0.0 : s;
      misc[0] += gv[k] * s, misc[1] += y * y, misc[2] -= ng * ng;
      const double os = __shfl_xor(s, X), ong = __shfl_xor(ng, X), oyf = __shfl_xor(yf, X), osa = __shfl_xor(sa, X);
      s4[2 * k] = hi ? os : s, s4[2 * k + 1] = hi ? s : os;
      ng4[2 * k] = hi ? ong : ng, ng4[2 * k + 1] = hi ? ng : ong;
      yf4[2 * k] = hi ? oyf : yf, yf4[2 * k + 1] = hi ? yf : oyf;
      sa4[2 * k] = hi ? osa : sa, sa4[2 * k + 1] = hi ? sa : osa;
    }
#pragma unroll
    for (int j = 0; j < H; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[0][j] = __builtin_fma(s4[k], a[j][k], acc[0][j]), acc[1][j] = __builtin_fma(b[j][k], s4[k], acc[1][j]);
        acc[2][j] = __builtin_fma(a[j][k], ng4[k], acc[2][j]), acc[3][j] = __builtin_fma(b[j][k], ng4[k], acc[3][j]);
        acc[4][j] = __builtin_fma(yf4[k], a[j][k], acc[4][j]), acc[5][j] = __builtin_fma(sa4[k], b[j][k], acc[5][j]);
        acc[6][j] = __builtin_fma(sa4[k], a[j][k], acc[6][j]), acc[7][j] = __builtin_fma(b[j][k], yf4[k], acc[7][j]);
      }
  }
  double tot = misc[0] + misc[1] + misc[2];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const int c = (hi ? H : 0) + j;
      if (c < NC) tot += acc[q][j] * (1.0 + 0.01 * (q * NC + c));
    }
  wave_sum_store(tot, sums);
}

// ---- update pass, compact, SLOT order: lane l works on the rows in slots l and l + 64 of the tile (T = 128 only) ----
// The W loads are dense: lanes [0, tf) of the first half read consecutive entries of the run, the second half is
// loaded only by tiles with more than 64 free rows (wave-uniform).  The row vectors are gathered through the slot ->
// row map of the tile (built per trip in LDS: one byte per slot).  Sums run in slot order.
__global__ __launch_bounds__(BLOCK) void upd_dense(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, double *sums) {
  __shared__ uint8_t perm[BLOCK / 64][128];
  const int64_t ntr = n / 128;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  UAcc A;
  uacc_zero(A);
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 128;
    const WaveTile wt = wave_tile(row0, L);
    const uint64_t below = (1ull << lane) - 1ull;
    const int c0 = __popcll(wt.m0), tf = c0 + __popcll(wt.m1);
    const int b0 = __popcll(wt.m0 & below), b1 = c0 + __popcll(wt.m1 & below);
    const int s0 = ((wt.m0 >> lane) & 1) ? b0 : tf + (lane - b0);
    const int s1 = ((wt.m1 >> lane) & 1) ? b1 : tf + (64 + lane - b1);
    perm[wv][s0] = (uint8_t)lane, perm[wv][s1] = (uint8_t)(64 + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t i0 = row0 + perm[wv][lane], i1 = row0 + perm[wv][64 + lane];
    __builtin_amdgcn_wave_barrier();
    const double xv[2] = {ldnt1(x + i0), ldnt1(x + i1)}, gv[2] = {ldnt1(g + i0), ldnt1(g + i1)};
    const double rv[2] = {ldnt1(r + i0), ldnt1(r + i1)}, tv[2] = {ldnt1(t + i0), ldnt1(t + i1)};
    const int fw[2] = {iw[i0], iw[i1]};
    double a[NC][2], b[NC][2];
    const bool n0 = lane < tf;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *pa = w + (int64_t)j * ld + row0, *pb = w + (int64_t)(NC + j) * ld + row0;
      a[j][0] = *(n0 ? pa + lane : zero), b[j][0] = *(n0 ? pb + lane : zero);
      a[j][1] = 0.0, b[j][1] = 0.0;
    }
    if (tf > 64) {  // (wave-uniform)
      const bool n1 = 64 + lane < tf;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const double *pa = w + (int64_t)j * ld + row0, *pb = w + (int64_t)(NC + j) * ld + row0;
        a[j][1] = *(n1 ? pa + 64 + lane : zero), b[j][1] = *(n1 ? pb + 64 + lane : zero);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double ak[NC], bk[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) ak[j] = a[j][k], bk[j] = b[j][k];
      urow(A, xv[k], gv[k], rv[k], tv[k], fw[k], ak, bk);
    }
  }
  uacc_out(A, sums);
}

// ---- SLOT order with 16-byte loads: tiles of 256 rows, lane l works on slots (2l, 2l + 1) and (128 + 2l, 129 + 2l) ----
// The free rows' W entries are the front of the tile: one dense 16-byte load per column and lane, the back pair only
// in tiles with more than 128 free rows (wave-uniform).  Row vectors: 8-byte gathers through the slot -> row map.
struct Tile256 {
  int tf;
  int rr[4];  // row (inside the tile) of this lane's four slots
};
__device__ __forceinline__ Tile256 tile256(const uint64_t *lmask, int64_t tr, uint8_t *perm, int lane) {
  uint64_t m[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) m[q] = lmask[4 * tr + q];
  int c[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) c[q] = __popcll(m[q]);
  Tile256 t;
  t.tf = c[0] + c[1] + c[2] + c[3];
  const uint64_t below = (1ull << lane) - 1ull;
  int run = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int b = run + __popcll(m[q] & below);
    const int row = 64 * q + lane;
    const int slot = ((m[q] >> lane) & 1) ? b : t.tf + (row - b);
    perm[slot] = (uint8_t)row;
    run += c[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uchar2 f = *reinterpret_cast<const uchar2 *>(perm + 2 * lane), bk = *reinterpret_cast<const uchar2 *>(perm + 128 + 2 * lane);
  __builtin_amdgcn_wave_barrier();
  t.rr[0] = f.x, t.rr[1] = f.y, t.rr[2] = bk.x, t.rr[3] = bk.y;
  return t;
}
__global__ __launch_bounds__(BLOCK) void upd_dense16(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, double *sums) {
  __shared__ uint8_t perm[BLOCK / 64][256];
  const int64_t ntr = n / 256;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  UAcc A;
  uacc_zero(A);
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 256;
    const Tile256 T = tile256(L.lmask, tr, perm[wv], lane);
    double xv[4], gv[4], rv[4], tv[4];
    int fw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = row0 + T.rr[k];
      xv[k] = ldnt1(x + i), gv[k] = ldnt1(g + i), rv[k] = ldnt1(r + i), tv[k] = ldnt1(t + i), fw[k] = iw[i];
    }
    d2 a[NC][2], b[NC][2];
    const bool n0 = 2 * lane < T.tf;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *pa = w + (int64_t)j * ld + row0 + 2 * lane, *pb = w + (int64_t)(NC + j) * ld + row0 + 2 * lane;
      a[j][0] = *reinterpret_cast<const d2 *>(n0 ? pa : zero), b[j][0] = *reinterpret_cast<const d2 *>(n0 ? pb : zero);
      a[j][1] = d2{0.0, 0.0}, b[j][1] = d2{0.0, 0.0};
    }
    if (T.tf > 128) {  // (wave-uniform)
      const bool n1 = 128 + 2 * lane < T.tf;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const double *pa = w + (int64_t)j * ld + row0 + 128 + 2 * lane, *pb = w + (int64_t)(NC + j) * ld + row0 + 128 + 2 * lane;
        a[j][1] = *reinterpret_cast<const d2 *>(n1 ? pa : zero), b[j][1] = *reinterpret_cast<const d2 *>(n1 ? pb : zero);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double ak[NC], bk[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) ak[j] = a[j][k >> 1][k & 1], bk[j] = b[j][k >> 1][k & 1];
      urow(A, xv[k], gv[k], rv[k], tv[k], fw[k], ak, bk);
    }
  }
  uacc_out(A, sums);
}
__global__ __launch_bounds__(BLOCK) void store_dense16(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, const double *__restrict__ zero, int64_t ld, Layout L, Coefs cf, double *xout, double *py, double *ps, double *sums) {
  __shared__ uint8_t perm[BLOCK / 64][256];
  const int64_t ntr = n / 256;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc = 0.0;
  TRIPS(tr, ntr) {
    const int64_t row0 = tr * 256;
    const Tile256 T = tile256(L.lmask, tr, perm[wv], lane);
    double xv[4], gv[4], rv[4], tv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = row0 + T.rr[k];
      xv[k] = ldnt1(x + i), gv[k] = ldnt1(g + i), rv[k] = ldnt1(r + i), tv[k] = ldnt1(t + i);
    }
    d2 a[NC][2], b[NC][2];
    const bool n0 = 2 * lane < T.tf;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *pa = w + (int64_t)j * ld + row0 + 2 * lane, *pb = w + (int64_t)(NC + j) * ld + row0 + 2 * lane;
      a[j][0] = *reinterpret_cast<const d2 *>(n0 ? pa : zero), b[j][0] = *reinterpret_cast<const d2 *>(n0 ? pb : zero);
      a[j][1] = d2{0.0, 0.0}, b[j][1] = d2{0.0, 0.0};
    }
    if (T.tf > 128) {
      const bool n1 = 128 + 2 * lane < T.tf;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const double *pa = w + (int64_t)j * ld + row0 + 128 + 2 * lane, *pb = w + (int64_t)(NC + j) * ld + row0 + 128 + 2 * lane;
        a[j][1] = *reinterpret_cast<const d2 *>(n1 ? pa : zero), b[j][1] = *reinterpret_cast<const d2 *>(n1 ? pb : zero);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    double z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int slot = (k >> 1) * 128 + 2 * lane + (k & 1);
      z[k] = xv[k];
      if (slot < T.tf) {
        double ak[NC], bk[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) ak[j] = a[j][k >> 1][k & 1], bk[j] = b[j][k >> 1][k & 1];
        z[k] = fmin(1.0, fmax(-1.0, xv[k] + newton_row(xv[k], gv[k], ak, bk, cf)));
      }
      const double dv = z[k] - xv[k];
      acc = acc + dv * gv[k];
      stnt1(xout + row0 + T.rr[k], z[k]);
    }
    stnt2(py + row0 + 2 * lane, d2{gv[0] - rv[0], gv[1] - rv[1]}), stnt2(py + row0 + 128 + 2 * lane, d2{gv[2] - rv[2], gv[3] - rv[3]});
    stnt2(ps + row0 + 2 * lane, d2{xv[0] - tv[0], xv[1] - tv[1]}), stnt2(ps + row0 + 128 + 2 * lane, d2{xv[2] - tv[2], xv[3] - tv[3]});
  }
  wave_sum_store(acc, sums);
}

// ---- ceiling of the access pattern: the update pass's bytes (tiles of 256: the front tf entries of 2 NC columns, all of
// x, g, r, t, iwhere) with 16-byte loads, one add per value, few registers (many waves) ----
__global__ __launch_bounds__(BLOCK) void ceil_pattern(int64_t n, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ r, const double *__restrict__ t, const int8_t *__restrict__ iw,
    const double *__restrict__ w, int64_t ld, Layout L, double *sums) {
  const int64_t ntr = n / 256;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  d2 acc = {0.0, 0.0};
  TRIPS(tr, ntr) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t grp = 2 * tr + h, row0 = grp * 128;
      const int64_t tbase = (row0 >> L.tshift) << L.tshift;
      const int git = (int)(row0 - tbase), tf = (int)L.ginfo[grp].y;   // (any tile size: the chunk [git, git + 128) of the tile's slots)
      const int64_t i = row0 + 2 * lane;
      acc += ldnt2(x + i) + ldnt2(g + i) + ldnt2(r + i) + ldnt2(t + i);
      const char2 c = *reinterpret_cast<const char2 *>(iw + i);
      acc[0] += c.x + c.y;
      if (git + 2 * lane < tf) {
#pragma unroll
        for (int j = 0; j < 2 * NC; ++j) acc += *reinterpret_cast<const d2 *>(w + (int64_t)j * ld + i);
      }
    }
  }
  wave_sum_store(acc[0] + acc[1], sums);
}
// for the check: the masked update kernel sees s = 0 on rows that are not free (as the real pass does: rows at a
// bound do not move) -- k_settle makes t = x there
__global__ void k_settle(int64_t n, const int8_t *iw, const double *x, double *t) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (iw[i] > 0) t[i] = x[i];
}

template <typename K>
static int resident_grid(K kern) {
  int per = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, BLOCK, 0));
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  return per * pr.multiProcessorCount;
}

template <typename F>
static double time_ms(F &&launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  double best = 1e30, tot = 0;
  for (int k = 0; k < reps; ++k) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min<double>(best, ms), tot += ms;
  }
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return tot / reps;
}

int main(int argc, char **argv) {
  const int64_t n = ((argc > 1 ? atoll(argv[1]) : 100000000ll) / 4096) * 4096;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int rot = argc > 3 ? atoi(argv[3]) : 0;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rot), &rot, sizeof rot));
  printf("trip rotation %d\n", rot);
  const int64_t ld = n;
  double *x, *g, *r, *t, *xout, *wn, *wc, *zero, *sums;
  int8_t *iw;
  uint64_t *lmask;
  uint2 *ginfo;
  const size_t vb = (size_t)n * 8;
  for (double **p : {&x, &g, &r, &t, &xout}) CK(hipMalloc(p, vb));
  CK(hipMalloc(&wn, vb * 2 * (NC + 1)));   // natural (+ the slot the pass stores its new pair into)
  CK(hipMalloc(&wc, vb * 2 * (NC + 1)));
  CK(hipMalloc(&zero, 256));
  CK(hipMemset(zero, 0, 256));
  CK(hipMalloc(&sums, 64));
  CK(hipMalloc(&iw, (size_t)n));
  CK(hipMalloc(&lmask, (size_t)(n / 64 + 2) * 8));
  CK(hipMalloc(&ginfo, (size_t)(n / 128 + 2) * sizeof(uint2)));
  Coefs cf;
  for (int j = 0; j < 2 * NC; ++j) cf.c[j] = 0.01 * (j + 1), cf.w[j] = -0.02 * (j + 2);
  const int g_sm = resident_grid(store_masked), g_sp = resident_grid(store_cpair), g_ss = resident_grid(store_csplit<false>);
  const int g_um = resident_grid(upd_masked), g_up = resident_grid(upd_compact<false>), g_us = resident_grid(upd_compact<true>);
  const int g_u2 = resident_grid(upd_pair2<1>), g_ud = resident_grid(upd_dense);
  printf("upd_dense resident grid %d\n", g_ud);
  printf("upd_pair2 resident grid %d\n", g_u2);
  printf("n = %lld rows, fp64, %d stored pairs + pending; resident grids: store %d / %d / %d, update %d / %d / %d\n",
         (long long)n, NC, g_sm, g_sp, g_ss, g_um, g_up, g_us);
  printf("%-5s %-6s %-6s | %-44s | %-44s\n", "free", "tile", "stale", "STORE pass ms (masked | c_pair | c_split)  B/row alg",
         "UPDATE pass ms (masked | c_pair | c_split)  B/row alg");
  auto get = [&]() { double h; CK(hipMemcpy(&h, sums, 8, hipMemcpyDeviceToHost)); return h; };
  for (double frac : {1.0, 0.5, 0.1}) {
    for (int T : {128, 256, 1024, 4096, 32768}) {
      for (int stale : {0, 1}) {
        if (stale && T != 1024) continue;
        if (T == 256 && frac == 0.1) continue;
        if (frac == 1.0 && T >= 4096) continue;
        int tshift = 0;
        while ((1 << tshift) < T) ++tshift;
        const uint32_t thresh = frac >= 1.0 ? 0xffffffffu : (uint32_t)(frac * 4294967296.0);
        k_fill_vec<<<2048, 256>>>(n, x, g, r, t);
        k_mask<<<(unsigned)((n / 64 + 255) / 256), 256>>>(n, thresh, stale ? (uint32_t)(0.005 * 4294967296.0) : 0u, lmask, iw);
        k_ginfo<<<(unsigned)((n / T + 255) / 256), 256>>>(n / 128, T / 128, lmask, ginfo);
        Layout L{lmask, ginfo, tshift};
        k_fill_w<<<4096, 256>>>(n, ld, wn, wc, L);
        k_settle<<<2048, 256>>>(n, iw, x, t);
        CK(hipDeviceSynchronize());
        // checksums
        double cs[6];
        CK(hipMemset(sums, 0, 64));
        store_masked<<<g_sm, BLOCK>>>(n, x, g, r, t, iw, wn, ld, cf, xout, wn + (int64_t)2 * NC * ld, wn + (int64_t)(2 * NC + 1) * ld, sums);
        cs[0] = get();
        CK(hipMemset(sums, 0, 64));
        store_cpair<<<g_sp, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums);
        cs[1] = get();
        CK(hipMemset(sums, 0, 64));
        store_csplit<false><<<g_ss, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums);
        cs[2] = get();
        CK(hipMemset(sums, 0, 64));
        upd_masked<<<g_um, BLOCK>>>(n, x, g, r, t, iw, wn, ld, sums);
        cs[3] = get();
        CK(hipMemset(sums, 0, 64));
        upd_compact<false><<<g_up, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums);
        cs[4] = get();
        CK(hipMemset(sums, 0, 64));
        upd_compact<true><<<g_us, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums);
        cs[5] = get();
        const bool ok = std::fabs(cs[1] - cs[0]) <= 1e-9 * std::fabs(cs[0]) && std::fabs(cs[2] - cs[0]) <= 1e-9 * std::fabs(cs[0]) &&
                        std::fabs(cs[4] - cs[3]) <= 1e-9 * std::fabs(cs[3]) && std::fabs(cs[5] - cs[3]) <= 1e-9 * std::fabs(cs[3]);
        // timings
        const double t_sm = time_ms([&] { store_masked<<<g_sm, BLOCK>>>(n, x, g, r, t, iw, wn, ld, cf, xout, wn + (int64_t)2 * NC * ld, wn + (int64_t)(2 * NC + 1) * ld, sums); }, reps);
        const double t_sp = time_ms([&] { store_cpair<<<g_sp, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums); }, reps);
        const double t_ss = time_ms([&] { store_csplit<false><<<g_ss, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums); }, reps);
        const double t_um = time_ms([&] { upd_masked<<<g_um, BLOCK>>>(n, x, g, r, t, iw, wn, ld, sums); }, reps);
        const double t_up = time_ms([&] { upd_compact<false><<<g_up, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums); }, reps);
        const double t_us = time_ms([&] { upd_compact<true><<<g_us, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums); }, reps);
        CK(hipMemset(sums, 0, 64));
        upd_pair2<1><<<g_u2, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums);
        const double cs_p2 = get();
        const double t_u2 = time_ms([&] { upd_pair2<1><<<g_u2, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums); }, reps);
        CK(hipMemset(sums, 0, 64));
        upd_pair2<32><<<g_u2, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums);
        const double cs_p32 = get();
        const double t_u32 = time_ms([&] { upd_pair2<32><<<g_u2, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums); }, reps);
        printf("      upd_pair2 (lane pairs share the column sums, tile trips): lanes (2p, 2p + 1) %6.3f ms  checksum %s | lanes (l, l + 32) %6.3f ms  checksum %s\n", t_u2,
               std::fabs(cs_p2 - cs[3]) <= 1e-9 * std::fabs(cs[3]) ? "ok" : "DIFFERS", t_u32,
               std::fabs(cs_p32 - cs[3]) <= 1e-9 * std::fabs(cs[3]) ? "ok" : "DIFFERS");
        if (T == 128) {
          static const int g_sd = resident_grid(store_dense<false>);
          CK(hipMemset(sums, 0, 64));
          store_dense<false><<<g_sd, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums);
          const double cs_sd = get();
          const double t_sd = time_ms([&] { store_dense<false><<<g_sd, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums); }, reps);
          printf("      store_dense (slot order, dense W loads, grid %d): %6.3f ms  checksum %s\n", g_sd, t_sd,
                 std::fabs(cs_sd - cs[0]) <= 1e-9 * std::fabs(cs[0]) ? "ok" : "DIFFERS");
          {
            static const int g_sl = resident_grid(store_csplit<true>), g_sdp = resident_grid(store_dense<true>);
            CK(hipMemset(sums, 0, 64));
            store_csplit<true><<<g_sl, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums);
            const double cs_l = get();
            const double t_l = time_ms([&] { store_csplit<true><<<g_sl, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums); }, reps);
            const double t_dp = time_ms([&] { store_dense<true><<<g_sdp, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums); }, reps);
            printf("      as the library loads W (by the bits, plain loads): c_split %6.3f ms (grid %d) checksum %s | slot order, dense %6.3f ms (grid %d)\n", t_l, g_sl,
                   std::fabs(cs_l - cs[0]) <= 1e-9 * std::fabs(cs[0]) ? "ok" : "DIFFERS", t_dp, g_sdp);
          }
          CK(hipMemset(sums, 0, 64));
          upd_dense<<<g_ud, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums);
          const double cs_d = get();
          const double t_ud = time_ms([&] { upd_dense<<<g_ud, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums); }, reps);
          printf("      upd_dense (slot order, dense W loads): %6.3f ms  checksum %s\n", t_ud,
                 std::fabs(cs_d - cs[3]) <= 1e-9 * std::fabs(cs[3]) ? "ok" : "DIFFERS");
        }
        {
          static const int g_cp = resident_grid(ceil_pattern);
          for (int mult : {2}) {
            const double t_cp = time_ms([&] { ceil_pattern<<<g_cp * mult / 2, BLOCK>>>(n, x, g, r, t, iw, wc, ld, L, sums); }, reps);
            printf("      ceil_pattern (the update pass's bytes, 16-byte loads, one add per value; grid %d): %6.3f ms = %4.2f TB/s\n", g_cp * mult / 2, t_cp,
                   (4 * 8 + 1 + 2 * NC * 8 * frac) * n / t_cp * 1e-9);
          }
        }
        if (T == 256) {
          static const int g_s16 = resident_grid(store_dense16), g_u16 = resident_grid(upd_dense16);
          CK(hipMemset(sums, 0, 64));
          store_dense16<<<g_s16, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums);
          const double cs_sd = get();
          const double t_sd = time_ms([&] { store_dense16<<<g_s16, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, cf, xout, wc + (int64_t)2 * NC * ld, wc + (int64_t)(2 * NC + 1) * ld, sums); }, reps);
          printf("      store_dense16 (slot order, 16-byte W loads, tiles of 256, grid %d): %6.3f ms  checksum %s\n", g_s16, t_sd,
                 std::fabs(cs_sd - cs[0]) <= 1e-9 * std::fabs(cs[0]) ? "ok" : "DIFFERS");
          CK(hipMemset(sums, 0, 64));
          upd_dense16<<<g_u16, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums);
          const double cs_d = get();
          const double t_ud = time_ms([&] { upd_dense16<<<g_u16, BLOCK>>>(n, x, g, r, t, iw, wc, zero, ld, L, sums); }, reps);
          printf("      upd_dense16 (slot order, 16-byte W loads, tiles of 256, grid %d): %6.3f ms  checksum %s\n", g_u16, t_ud,
                 std::fabs(cs_d - cs[3]) <= 1e-9 * std::fabs(cs[3]) ? "ok" : "DIFFERS");
        }
        const double nf = frac;  // (stale: +- 0.5 %)
        const double b_sm = 4 * 8 + 1 + 2 * NC * 8 + 3 * 8, b_sc = 4 * 8 + 1 + 2 * NC * 8 * nf + 3 * 8 + 0.19;
        const double b_um = 4 * 8 + 1 + 2 * NC * 8, b_uc = 4 * 8 + 1 + 2 * NC * 8 * nf + 0.19;
        printf("%-5.2f %-6d %-6s | %6.3f | %6.3f | %6.3f   %5.1f -> %5.1f (%4.2f TB/s -> %4.2f) | %6.3f | %6.3f | %6.3f   %5.1f -> %5.1f (%4.2f TB/s -> %4.2f) %s\n",
               frac, T, stale ? "0.5%" : "-", t_sm, t_sp, t_ss, b_sm, b_sc, b_sm * n / t_sm * 1e-9, b_sc * n / std::min(t_sp, t_ss) * 1e-9,
               t_um, t_up, t_us, b_um, b_uc, b_um * n / t_um * 1e-9, b_uc * n / std::min(t_up, t_us) * 1e-9, ok ? "sums ok" : "SUMS DIFFER");
        if (!ok) printf("   checksums: %.12e %.12e %.12e | %.12e %.12e %.12e\n", cs[0], cs[1], cs[2], cs[3], cs[4], cs[5]);
        fflush(stdout);
      }
    }
  }
  return 0;
}
