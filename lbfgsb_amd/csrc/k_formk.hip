// k_formk.hip -- formk: masked Gram of W from scratch and sparse patches
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

namespace lbk {

// =========================== formk inner products ============================
// From-scratch masked Gram of [Wy Ws] (reference keeps wn1 incrementally,
// :1735-1851; same sums, same row sets, no cliff when many variables change
// status).  Row tiles are staged in LDS once and every needed product pair reads
// them from there; each output entry is owned by exactly one lane of the
// workgroup, so no cross-lane reduction is needed.
template <int MC>
struct GramCfg {
  static constexpr int R = MC <= 20 ? 128 : 64;     // rows per tile
  static constexpr int RS = 2 * MC + 1;             // LDS row stride (odd: spreads banks)
  static constexpr int E = 2 * MC * MC + MC;        // outputs at col == MC
  static constexpr int NE = (E + BLOCK - 1) / BLOCK;  // outputs per lane
};
template <typename T, int MC>
__global__ __launch_bounds__(BLOCK) void formk_gram_kernel(int64_t n, const T *__restrict__ ws,
                                                           const T *__restrict__ wy, int64_t ldw,
                                                           int m, int head, int col,
                                                           const iw_t *__restrict__ iwhere,
                                                           double *gpart) {
  using C = GramCfg<MC>;
  __shared__ double tile[C::R * C::RS];
  __shared__ int flag[C::R];
  const int tri = col * (col + 1) / 2;
  const int E = 2 * col * col + col;
  // which (column a, column b, row set) this lane owns
  int ca[C::NE], cb[C::NE], want[C::NE];
  double acc[C::NE];
#pragma unroll
  for (int s = 0; s < C::NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    acc[s] = 0.0;
    ca[s] = cb[s] = 0;
    want[s] = 2;  // matches no row
    if (e < E) {
      if (e < 2 * tri) {
        const int ee = e < tri ? e : e - tri;
        int i = (int)((sqrt(8.0 * ee + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > ee) --i;
        while ((i + 1) * (i + 2) / 2 <= ee) ++i;
        const int j = ee - i * (i + 1) / 2;
        if (e < tri) {
          ca[s] = i;
          cb[s] = j;
          want[s] = 1;  // free rows: Wy_i . Wy_j
        } else {
          ca[s] = col + i;
          cb[s] = col + j;
          want[s] = 0;  // active rows: Ws_i . Ws_j
        }
      } else {
        const int ee = e - 2 * tri;
        const int i = ee / col, j = ee % col;
        ca[s] = col + i;  // Ws_i
        cb[s] = j;        // Wy_j
        want[s] = i > j ? 0 : 1;
      }
    }
  }
  const int64_t ntiles = (n + C::R - 1) / C::R;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t r0 = t * C::R;
    __syncthreads();
    for (int qd = threadIdx.x; qd < 2 * col * C::R; qd += BLOCK) {
      const int c = qd / C::R, r = qd % C::R;
      const int64_t row = r0 + r;
      double v = 0.0;
      if (row < n) {
        const int jj = c < col ? c : c - col;
        const int64_t off = (int64_t)((head - 1 + jj) % m) * ldw + row;
        v = c < col ? (double)wy[off] : (double)ws[off];
      }
      tile[r * C::RS + c] = v;
    }
    for (int r = threadIdx.x; r < C::R; r += BLOCK) {
      const int64_t row = r0 + r;
      flag[r] = row < n ? (iwhere[row] <= 0 ? 1 : 0) : 3;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < C::R; ++r) {
      const int f = flag[r];
#pragma unroll
      for (int s = 0; s < C::NE; ++s) {
        const double a = tile[r * C::RS + ca[s]];
        const double b = tile[r * C::RS + cb[s]];
        if (f == want[s]) acc[s] += a * b;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < C::NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    if (e < E) gpart[(size_t)e * GRAM_BLOCKS + blockIdx.x] = acc[s];
  }
}
// Row-parallel variant for col <= 10 (the benchmark's m = 10).  A 512-thread workgroup
// (8 waves) takes a 128-row slab: each wave loads 1/8 of the 2*MC columns once (16 B per
// lane, coalesced) into a double-buffered LDS slab (conflict-free 16-byte slots, ONE barrier
// per slab, the next slab's global loads in flight during the math), and each wave owns one
// eighth of the outputs in registers -- matrix = wave/2: Y'ZZ'Y (free rows), S'AA'S (active
// rows), R_z (free, i<=j), L_a (active, i>j); half = wave%2 splits the outer index at H.
// <= 28 accumulators per lane keep it under 128 VGPRs: two workgroups (16 waves) per CU.
// HBM traffic is exactly one pass over W plus iwhere.
template <int MC>
struct GramRows {
  static constexpr int H = MC == 10 ? 7 : (MC + 1) / 2 + 1;  // outer-index split
  static constexpr int NACC = H * (H + 1) / 2 > (MC - H) * (MC + H + 1) / 2
                                  ? H * (H + 1) / 2
                                  : (MC - H) * (MC + H + 1) / 2;
  static constexpr int ROWS = 128;  // 64 lanes x 2 rows
  static constexpr int NW = 8;
};

// accumulate one slab for role (MT, HALF); a = pointer to the slab [2*MC][64] of double2
template <int MC, int MT, int HALF>
__device__ __forceinline__ void gram_role(const double2 (*__restrict__ sl)[64], int lane, double m0,
                                          double m1, double (&acc)[GramRows<MC>::NACC]) {
  constexpr int H = GramRows<MC>::H;
  constexpr int LO = HALF == 0 ? 0 : H, HI = HALF == 0 ? H : MC;
  // inner operands are re-read from LDS (cheap: the LDS pipe is otherwise idle)
  if constexpr (MT == 0 || MT == 1) {
    constexpr int C0 = MT == 0 ? 0 : MC;  // Y block or S block
    int k = 0;
#pragma unroll
    for (int i = LO; i < HI; ++i) {
      const double2 ai = sl[C0 + i][lane];
      const double ax = ai.x * m0, ay = ai.y * m1;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const double2 aj = sl[C0 + j][lane];
        // (a reduction, summed in another order than the reference's anyway: fused multiply-adds
        //  halve the fp64 issue slots of this compute-heavy pass -- 420 flops per row)
        acc[k] = __builtin_fma(ax, aj.x, __builtin_fma(ay, aj.y, acc[k]));
        ++k;
      }
    }
  } else if constexpr (MT == 2) {  // R_z: Ws_i . Wy_j, free rows, i <= j, outer j in [LO,HI)
    int k = 0;
#pragma unroll
    for (int j = LO; j < HI; ++j) {
      const double2 y = sl[j][lane];
      const double yx = y.x * m0, yy = y.y * m1;
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        const double2 sv = sl[MC + i][lane];
        acc[k] = __builtin_fma(sv.x, yx, __builtin_fma(sv.y, yy, acc[k]));
        ++k;
      }
    }
  } else {  // L_a: Ws_i . Wy_j, active rows, i > j, outer i in [max(LO,1),HI)
    int k = 0;
#pragma unroll
    for (int i = (LO < 1 ? 1 : LO); i < HI; ++i) {
      const double2 sv = sl[MC + i][lane];
      const double sx = sv.x * m0, sy = sv.y * m1;
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const double2 y = sl[j][lane];
        acc[k] = __builtin_fma(sx, y.x, __builtin_fma(sy, y.y, acc[k]));
        ++k;
      }
    }
  }
}
// write one role's outputs (same enumeration order as gram_role)
template <int MC, int MT, int HALF>
__device__ __forceinline__ void gram_store(const double (&acc)[GramRows<MC>::NACC], int lane, int col,
                                           double *gpart) {
  constexpr int H = GramRows<MC>::H;
  constexpr int LO = HALF == 0 ? 0 : H, HI = HALF == 0 ? H : MC;
  const int tri = col * (col + 1) / 2;
  int k = 0;
  if constexpr (MT == 0 || MT == 1) {
#pragma unroll
    for (int i = LO; i < HI; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const double v = wave_sum(acc[k++]);
        if (lane == 0 && i < col)
          gpart[(size_t)((MT == 0 ? 0 : tri) + i * (i + 1) / 2 + j) * GRAM_BLOCKS + blockIdx.x] = v;
      }
  } else if constexpr (MT == 2) {
#pragma unroll
    for (int j = LO; j < HI; ++j)
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        const double v = wave_sum(acc[k++]);
        if (lane == 0 && j < col) gpart[(size_t)(2 * tri + i * col + j) * GRAM_BLOCKS + blockIdx.x] = v;
      }
  } else {
#pragma unroll
    for (int i = (LO < 1 ? 1 : LO); i < HI; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const double v = wave_sum(acc[k++]);
        if (lane == 0 && i < col) gpart[(size_t)(2 * tri + i * col + j) * GRAM_BLOCKS + blockIdx.x] = v;
      }
  }
}

template <typename T, int MC>
__global__ __launch_bounds__(512) void formk_gram_rows_kernel(
    int64_t n, const T *__restrict__ ws, const T *__restrict__ wy, int64_t ldw, int m, int head,
    int col, const iw_t *__restrict__ iwhere, double *gpart) {
  using G = GramRows<MC>;
  constexpr int NC = 2 * MC;                      // columns: [0,MC) = Wy, [MC,2MC) = Ws
  constexpr int PER = (NC + G::NW - 1) / G::NW;   // columns loaded per wave
  __shared__ double2 slab[2][NC][64];
  __shared__ int2 fl[2][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform role
  double acc[G::NACC];
#pragma unroll
  for (int k = 0; k < G::NACC; ++k) acc[k] = 0.0;

  const int64_t nslab = (n + G::ROWS - 1) / G::ROWS;
  // two register stages: while slab t is computed from LDS, slabs t+1 and t+2 are in flight
  double2 stA[PER], stB[PER];
  int2 fA = make_int2(3, 3), fB = make_int2(3, 3);
  auto issue = [&](int64_t sl, double2(&stage)[PER], int2 &fstage) {
    const int64_t r0 = sl * G::ROWS + 2 * lane;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = w + G::NW * q;
      double2 v = make_double2(0.0, 0.0);
      if (sl < nslab && c < NC && r0 < n) {
        const int j = c < MC ? c : c - MC;
        const T *base = (c < MC ? wy : ws) + col_off(j, col, head, m, ldw) + r0;
        if (r0 + 1 < n) {
          double t2[2];
          ld<2>(base, t2);
          v = make_double2(t2[0], t2[1]);
        } else {
          v.x = (double)base[0];
        }
      }
      stage[q] = v;
    }
    if (w == G::NW - 1) {
      fstage = make_int2(3, 3);
      if (sl < nslab && r0 < n) fstage.x = iwhere[r0] <= 0 ? 1 : 0;
      if (sl < nslab && r0 + 1 < n) fstage.y = iwhere[r0 + 1] <= 0 ? 1 : 0;
    }
  };
  auto put = [&](int buf, const double2(&stage)[PER], const int2 &fstage) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = w + G::NW * q;
      if (c < NC) slab[buf][c][lane] = stage[q];
    }
    if (w == G::NW - 1) fl[buf][lane] = fstage;
  };
  auto math = [&](int buf) {
    const int2 f = fl[buf][lane];
    const int mt = w >> 1;
    const int want = (mt == 0 || mt == 2) ? 1 : 0;  // free rows for Y'ZZ'Y and R_z
    const double m0 = f.x == want ? 1.0 : 0.0, m1 = f.y == want ? 1.0 : 0.0;
    switch (w) {
      case 0: gram_role<MC, 0, 0>(slab[buf], lane, m0, m1, acc); break;
      case 1: gram_role<MC, 0, 1>(slab[buf], lane, m0, m1, acc); break;
      case 2: gram_role<MC, 1, 0>(slab[buf], lane, m0, m1, acc); break;
      case 3: gram_role<MC, 1, 1>(slab[buf], lane, m0, m1, acc); break;
      case 4: gram_role<MC, 2, 0>(slab[buf], lane, m0, m1, acc); break;
      case 5: gram_role<MC, 2, 1>(slab[buf], lane, m0, m1, acc); break;
      case 6: gram_role<MC, 3, 0>(slab[buf], lane, m0, m1, acc); break;
      default: gram_role<MC, 3, 1>(slab[buf], lane, m0, m1, acc); break;
    }
  };
  const int64_t g = gridDim.x;
  issue(blockIdx.x, stA, fA);
  issue(blockIdx.x + g, stB, fB);
  for (int64_t sl = blockIdx.x; sl < nslab; sl += 2 * g) {
    put(0, stA, fA);
    __syncthreads();
    issue(sl + 2 * g, stA, fA);
    math(0);
    if (sl + g < nslab) {  // uniform over the workgroup
      put(1, stB, fB);
      __syncthreads();
      issue(sl + 3 * g, stB, fB);
      math(1);
    }
  }
  switch (w) {
    case 0: gram_store<MC, 0, 0>(acc, lane, col, gpart); break;
    case 1: gram_store<MC, 0, 1>(acc, lane, col, gpart); break;
    case 2: gram_store<MC, 1, 0>(acc, lane, col, gpart); break;
    case 3: gram_store<MC, 1, 1>(acc, lane, col, gpart); break;
    case 4: gram_store<MC, 2, 0>(acc, lane, col, gpart); break;
    case 5: gram_store<MC, 2, 1>(acc, lane, col, gpart); break;
    case 6: gram_store<MC, 3, 0>(acc, lane, col, gpart); break;
    default: gram_store<MC, 3, 1>(acc, lane, col, gpart); break;
  }
}

// Quad variant (col <= 10): no LDS, no barriers.  The four lanes of a quad share the rows of all
// four: lane q = lane & 3 owns ONE of the four masked triangles of WN1 --
//   q = 0: Y'ZZ'Y (free rows), q = 1: S'AA'S (active rows),
//   q = 2: R_z = sum_free S_i Y_j for i <= j, q = 3: L_a = sum_active S_i Y_j for i > j --
// MC(MC+1)/2 fp64 sums per lane instead of 2 MC^2 + MC, and sees every row of its quad through
// DPP quad permutes (a VALU move, no LDS).  All four triangles have the shape
// acc[a, b] += U_a V_b for a >= b with U, V in {Y, S} of the row, so one instruction stream
// serves the four roles: U and V are selected per lane.  Loads are 8 bytes per lane and column
// (one fp64 row / two fp32 rows), straight to registers; the whole trip is issued before its
// first use.  HBM traffic: one pass over W plus iwhere.
template <typename T>
struct GramQuad {
  static constexpr int W = sizeof(T) == 8 ? 1 : 2;  // rows per lane
};
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);  // (quad_perm only: see pair_xchg)
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = dpp_i<CTRL>((int)b), hi = dpp_i<CTRL>((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(float v) {  // the fp32 operand travels, widened on arrival
  return (double)__builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v)));
}
template <typename T, int MC>
__global__ __launch_bounds__(BLOCK) void formk_gram_quad_kernel(
    int64_t n, const T *__restrict__ ws, const T *__restrict__ wy, const T *__restrict__ zero,
    int64_t ldw, int m, int head, int col, const iw_t *__restrict__ iwhere, double *gpart) {
  constexpr int W = GramQuad<T>::W, TRI = MC * (MC + 1) / 2;
  double acc[TRI];
#pragma unroll
  for (int k = 0; k < TRI; ++k) acc[k] = 0.0;
  const int lane = threadIdx.x & 63, q = lane & 3;
  const bool u_is_s = q & 1, v_is_s = q == 1 || q == 2, want_free = (q & 1) == 0;
  // every lane of a quad must run the same number of trips: groups of 4 * W rows, the ragged
  // end is padded with rows that read the zero buffer and count as neither free nor active
  const int64_t ngroups = (n + 4 * W - 1) / (4 * W);
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x / 4);
  for (int64_t gq = (int64_t)blockIdx.x * (blockDim.x / 4) + (threadIdx.x >> 2); gq < ngroups; gq += stride) {
    const int64_t i0 = (gq * 4 + q) * W;  // this lane's first row
    T y[MC][W], sv[MC][W];
    int fl[W];
    const bool in = i0 + W <= n;  // (partial groups: element-wise below)
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      const int64_t off = col_off(j, col, head, m, ldw) + i0;
      const bool live = j < col && in;
      if constexpr (W == 1) {
        y[j][0] = __builtin_nontemporal_load(live ? wy + off : zero);
        sv[j][0] = __builtin_nontemporal_load(live ? ws + off : zero);
      } else {
        typedef T t2 __attribute__((ext_vector_type(2)));
        const t2 a = __builtin_nontemporal_load(reinterpret_cast<const t2 *>(live ? wy + off : zero));
        const t2 b = __builtin_nontemporal_load(reinterpret_cast<const t2 *>(live ? ws + off : zero));
        y[j][0] = a.x, y[j][1] = a.y, sv[j][0] = b.x, sv[j][1] = b.y;
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) fl[k] = in ? (iwhere[i0 + k] <= 0 ? 1 : 0) : 2;
    if (!in) {  // ragged end: the rows that exist, one by one
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if (i0 + k < n) {
          fl[k] = iwhere[i0 + k] <= 0 ? 1 : 0;
#pragma unroll
          for (int j = 0; j < MC; ++j) {
            if (j < col) {
              const int64_t off = col_off(j, col, head, m, ldw) + i0 + k;
              y[j][k] = wy[off], sv[j][k] = ws[off];
            }
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    auto add_row = [&](const double (&Y)[MC], const double (&S)[MC], int f) {
      const double mk = (f == (want_free ? 1 : 0)) ? 1.0 : 0.0;
      double U[MC], V[MC];
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        U[c] = (u_is_s ? S[c] : Y[c]) * mk;
        V[c] = v_is_s ? S[c] : Y[c];
      }
      int t = 0;
#pragma unroll
      for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          acc[t] = __builtin_fma(U[a], V[b], acc[t]);
          ++t;
        }
    };
#pragma unroll
    for (int k = 0; k < W; ++k) {
      double Y[MC], S[MC];
#pragma unroll
      for (int c = 0; c < MC; ++c) Y[c] = (double)y[c][k], S[c] = (double)sv[c][k];
      add_row(Y, S, fl[k]);
#define LB_MATE(CTRL)                                                     \
      {                                                                   \
        _Pragma("unroll") for (int c = 0; c < MC; ++c) {                  \
          Y[c] = dpp_d<CTRL>(y[c][k]);                                    \
          S[c] = dpp_d<CTRL>(sv[c][k]);                                   \
        }                                                                 \
        add_row(Y, S, dpp_i<CTRL>(fl[k]));                                \
      }
      LB_MATE(0xB1)  // lane ^ 1
      LB_MATE(0x4E)  // lane ^ 2
      LB_MATE(0x1B)  // lane ^ 3
#undef LB_MATE
    }
  }
  // lanes with equal q hold the same triangle: reduce over them, then across the 4 waves
  __shared__ double sm[4][4][TRI];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < TRI; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o);
    if (lane < 4) sm[w][lane][k] = v;
  }
  __syncthreads();
  const int tri = col * (col + 1) / 2;
  for (int e = threadIdx.x; e < 4 * TRI; e += blockDim.x) {
    const int role = e / TRI, k = e % TRI;
    int a = (int)((sqrt(8.0 * k + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > k) --a;
    while ((a + 1) * (a + 2) / 2 <= k) ++a;
    const int b = k - a * (a + 1) / 2;
    if (a >= col) continue;
    const double sum = ((sm[0][role][k] + sm[1][role][k]) + sm[2][role][k]) + sm[3][role][k];
    int slot;
    if (role == 0)
      slot = a * (a + 1) / 2 + b;             // Y'ZZ'Y (i = a, j = b)
    else if (role == 1)
      slot = tri + a * (a + 1) / 2 + b;       // S'AA'S
    else if (role == 2)
      slot = 2 * tri + b * col + a;           // R_z: is = b <= jy = a
    else {
      if (a == b) continue;                   // L_a has no diagonal
      slot = 2 * tri + a * col + b;           // L_a: is = a > jy = b
    }
    gpart[(size_t)slot * GRAM_BLOCKS + blockIdx.x] = sum;
  }
}

template <typename T>
void launch_formk_gram(Queue &q, int64_t n, WStore<T> w, int head, int col,
                       const iw_t *iwhere) {
  int gr = 0;
  if (col <= 10) {
    const int variant = q.tune.gram_rows;  // 1: the LDS-slab kernel (A/B timing)
    if (variant == 1) {
      const int64_t nslab = (n + 127) / 128;
      gr = (int)(nslab < GRAM_BLOCKS ? nslab : GRAM_BLOCKS);
      if (col <= 5)
        hipLaunchKernelGGL((formk_gram_rows_kernel<T, 5>), dim3(gr), dim3(512), 0, q.stream, n, w.ws,
                           w.wy, w.ld, w.m, head, col, iwhere, q.d_gpart);
      else
        hipLaunchKernelGGL((formk_gram_rows_kernel<T, 10>), dim3(gr), dim3(512), 0, q.stream, n,
                           w.ws, w.wy, w.ld, w.m, head, col, iwhere, q.d_gpart);
    } else {
      const int64_t ngroups = (n + 4 * GramQuad<T>::W - 1) / (4 * GramQuad<T>::W);
      const int64_t want = (ngroups + BLOCK / 4 - 1) / (BLOCK / 4);
      gr = (int)(want < GRAM_BLOCKS ? want : GRAM_BLOCKS);
      if (col <= 5)
        hipLaunchKernelGGL((formk_gram_quad_kernel<T, 5>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws,
                           w.wy, w.zero, w.ld, w.m, head, col, iwhere, q.d_gpart);
      else
        hipLaunchKernelGGL((formk_gram_quad_kernel<T, 10>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws,
                           w.wy, w.zero, w.ld, w.m, head, col, iwhere, q.d_gpart);
    }
    LB_LAUNCHED(q);
    finalize_from(q, q.d_gpart, GRAM_BLOCKS, gr, 2 * col * col + col, 0, 0);
    return;
  }
  DISPATCH_MAXC(col, {
    const int64_t ntiles = (n + GramCfg<MC>::R - 1) / GramCfg<MC>::R;
    gr = (int)(ntiles < GRAM_BLOCKS ? ntiles : GRAM_BLOCKS);
    hipLaunchKernelGGL((formk_gram_kernel<T, MC>), dim3(gr), dim3(BLOCK), 0, q.stream, n, w.ws,
                       w.wy, w.ld, w.m, head, col, iwhere, q.d_gpart);
  });
  LB_LAUNCHED(q);
  finalize_from(q, q.d_gpart, GRAM_BLOCKS, gr, 2 * col * col + col, 0, 0);
}

// formk's patches for variables that changed status (:1801-1851): signed Gram over the
// listed rows only, sign +1 for rows that entered the free set, -1 for rows that left it.
// chg[k] = local row | (left ? 0x80000000 : 0).  Output layout = the Gram's (E entries for
// `upcl` columns): P_yy (i>=j), P_ss (i>=j), P_sy (all i,j).
template <typename T>
// cnt_ptr != nullptr: the length of the list is read from the device (the eager chain that follows freev's
// counting pass without a host round trip); a list longer than `cnt` (= the capacity the chain serves) is not
// patched at all and slot E of the output says so (1.0), so that the host -- every rank's -- takes the
// ordinary route.
__global__ __launch_bounds__(BLOCK) void formk_patch_kernel(const uint32_t *__restrict__ chg,
                                                            uint32_t cnt, const uint32_t *cnt_ptr,
                                                            const T *__restrict__ ws,
                                                            const T *__restrict__ wy, int64_t ldw,
                                                            int m, int head, int upcl,
                                                            double *gpart,
                                                            const uint64_t *__restrict__ lmask) {
  constexpr int R = 64, RS = 2 * MAXM + 1;
  __shared__ double tile[R * RS];
  __shared__ double sgn[R];
  const int tri = upcl * (upcl + 1) / 2;
  const int E = 2 * upcl * upcl + upcl;
  if (cnt_ptr) {
    const uint32_t have = *cnt_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) gpart[(size_t)E * GRAM_BLOCKS] = have > cnt ? 1.0 : 0.0;
    if (blockIdx.x > 0 && threadIdx.x == 0) gpart[(size_t)E * GRAM_BLOCKS + blockIdx.x] = 0.0;
    cnt = have > cnt ? 0u : have;
  }
  constexpr int NE = (2 * MAXM * MAXM + MAXM + BLOCK - 1) / BLOCK;
  int ca[NE], cb[NE];
  double acc[NE];
#pragma unroll
  for (int s = 0; s < NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    acc[s] = 0.0;
    ca[s] = cb[s] = 0;
    if (e < E) {
      if (e < 2 * tri) {
        const int ee = e < tri ? e : e - tri;
        int i = (int)((sqrt(8.0 * ee + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > ee) --i;
        while ((i + 1) * (i + 2) / 2 <= ee) ++i;
        const int j = ee - i * (i + 1) / 2;
        ca[s] = (e < tri ? 0 : upcl) + i;
        cb[s] = (e < tri ? 0 : upcl) + j;
      } else {
        const int ee = e - 2 * tri;
        ca[s] = upcl + ee / upcl;  // Ws_i
        cb[s] = ee % upcl;         // Wy_j
      }
    }
  }
  const uint32_t ntile = (cnt + R - 1) / R;
  for (uint32_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    __syncthreads();
    for (int qd = threadIdx.x; qd < 2 * upcl * R; qd += BLOCK) {
      const int c = qd / R, rr = qd % R;
      const uint32_t k = t * R + rr;
      double v = 0.0;
      if (k < cnt) {
        const int64_t row = chg[k] & 0x7FFFFFFFu;
        const int jj = c < upcl ? c : c - upcl;
        const int64_t off = (int64_t)((head - 1 + jj) % m) * ldw + wrow(lmask, row);
        v = c < upcl ? (double)wy[off] : (double)ws[off];
      }
      tile[rr * RS + c] = v;
    }
    for (int rr = threadIdx.x; rr < R; rr += BLOCK) {
      const uint32_t k = t * R + rr;
      sgn[rr] = k < cnt ? ((chg[k] & 0x80000000u) ? -1.0 : 1.0) : 0.0;
    }
    __syncthreads();
    for (int rr = 0; rr < R; ++rr) {
      const double sg = sgn[rr];
#pragma unroll
      for (int s = 0; s < NE; ++s) acc[s] += sg * tile[rr * RS + ca[s]] * tile[rr * RS + cb[s]];
    }
  }
#pragma unroll
  for (int s = 0; s < NE; ++s) {
    const int e = threadIdx.x + s * BLOCK;
    if (e < E) gpart[(size_t)e * GRAM_BLOCKS + blockIdx.x] = acc[s];
  }
}
template <typename T>
void launch_formk_patch(Queue &q, const uint32_t *chg, uint32_t cnt, WStore<T> w, int head, int upcl) {
  int gr = (int)((cnt + 63) / 64);
  if (gr < 1) gr = 1;
  if (gr > 256) gr = 256;
  hipLaunchKernelGGL(formk_patch_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, chg, cnt,
                     (const uint32_t *)nullptr, w.ws, w.wy, w.ld, w.m, head, upcl, q.d_gpart, w.lmask);
  LB_LAUNCHED(q);
  finalize_from(q, q.d_gpart, GRAM_BLOCKS, gr, 2 * upcl * upcl + upcl, 0, 0);
}
// the eager form: list length on the device, at most `cap` rows served (a fixed grid of cap / 64 workgroups:
// one tile each, the same association of the sums as the ordinary launch for a list of that length);
// E + 1 sums: the patch, then the "not served" flag
template <typename T>
void launch_formk_patch_dev(Queue &q, const uint32_t *chg, const uint32_t *cnt_ptr, uint32_t cap, WStore<T> w,
                            int head, int upcl) {
  const int gr = (int)((cap + 63) / 64);
  hipLaunchKernelGGL(formk_patch_kernel<T>, dim3(gr), dim3(BLOCK), 0, q.stream, chg, cap, cnt_ptr, w.ws, w.wy,
                     w.ld, w.m, head, upcl, q.d_gpart, w.lmask);
  LB_LAUNCHED(q);
  finalize_from(q, q.d_gpart, GRAM_BLOCKS, gr, 2 * upcl * upcl + upcl + 1, 0, 0);
}

// =========================== explicit instantiations =========================
#define INSTANTIATE(T) \
  template void launch_formk_gram<T>(Queue &, int64_t, WStore<T>, int, int, const iw_t *); \
  template void launch_formk_patch<T>(Queue &, const uint32_t *, uint32_t, WStore<T>, int, int); \
  template void launch_formk_patch_dev<T>(Queue &, const uint32_t *, const uint32_t *, uint32_t, WStore<T>, int, int);
INSTANTIATE(double)
INSTANTIATE(float)
#undef INSTANTIATE

}  // namespace lbk
