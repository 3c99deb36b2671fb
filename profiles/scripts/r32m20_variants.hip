// Micro-benchmark: variants of the REAL32, m = 20 cmprlb + W'r + formk-new-row pass
// (BASELINE.json configs[4]; cmprlb_wtv_kernel<float, 20, NEWROW> of the library ran one wave per
// SIMD at 3.7 TB/s in round 1).  40 fp32 columns of W + x, g (fp32) + iwhere (1 byte) per row,
// 120 fp64 sums.  All variants compute the same sums (checked against variant 0).
//   v0  as shipped: operands widened to fp64 on load, 2 rows per lane, 120 accumulators per lane
//   v1  operands stay fp32 in registers, widened where used
//   v2  v1 + the next trip's operands are loaded while this trip is summed (ping-pong registers)
//   v3  v1 with 4 rows per lane (16-byte loads)
//   v5  two neighbouring lanes share the accumulators (60 each), operands exchanged by DPP
//   v4  four neighbouring lanes share the accumulators (30 each), operands exchanged by DPP
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off r32m20_variants.hip -o r32m20
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int MC = 20;
constexpr int NA = 6 * MC;
constexpr int MAXB = 2048;
struct Coef { double a[2 * MC]; };
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W>
__device__ __forceinline__ void ldraw(const float *p, float (&o)[W]) {
  if constexpr (W == 2) {
    const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p));
    o[0] = v.x, o[1] = v.y;
  } else {
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
  }
}
template <int W>
__device__ __forceinline__ void ldiw(const int8_t *p, int (&o)[W]) {
  if constexpr (W == 2) {
    const char2 v = *reinterpret_cast<const char2 *>(p);
    o[0] = v.x, o[1] = v.y;
  } else {
    const char4 v = *reinterpret_cast<const char4 *>(p);
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
  }
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}
template <int K>
__device__ __forceinline__ void block_store(const double (&acc)[K], double *part) {
  __shared__ double sm[4][K];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) sm[w][k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x)
    part[(size_t)k * MAXB + blockIdx.x] = ((sm[0][k] + sm[1][k]) + sm[2][k]) + sm[3][k];
}
__device__ __forceinline__ double widen_late(float v) {
  double d;
  asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(v));
  return d;
}
// r of one row (cmprlb :1565-1583) from raw operands; the new pair is logical column MC-1
template <int W, typename A>
__device__ __forceinline__ void row_terms(const A &a, const A &b, const float (&xv)[W], const float (&gv)[W],
                                          const int (&iw)[W], double tsum, double theta, const Coef &cf,
                                          double (&rv)[W], double (&yf)[W], double (&sa)[W]) {
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const double xk = xv[k], gk = gv[k];
    const double zk = (iw[k] == 0 || iw[k] == -1) ? (double)(float)(xk + tsum * (-gk)) : xk;
    double rr = -theta * (zk - xk) - gk;
#pragma unroll
    for (int j = 0; j < MC; ++j) rr = rr + (double)a[j][k] * cf.a[j] + (double)b[j][k] * cf.a[MC + j];
    rv[k] = iw[k] <= 0 ? rr : 0.0;
    yf[k] = iw[k] <= 0 ? (double)a[MC - 1][k] : 0.0;
    sa[k] = iw[k] <= 0 ? 0.0 : (double)b[MC - 1][k];
  }
}

struct Args {
  int64_t n;
  const float *x, *g, *ws, *wy;
  const int8_t *iw;
  int64_t ld;
  double tsum, theta;
  Coef cf;
  double *part;
};

// ---------------------------------------------------------------- v0: shipped shape
__global__ __launch_bounds__(256) void v0(Args p) {
  constexpr int W = 2;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const int64_t nv = p.n / W, stride = (int64_t)gridDim.x * 256;
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    const int64_t i = iv * W;
    float xr[W], gr[W], t[W];
    double a[MC][W], b[MC][W], rv[W], yf[W], sa[W];
    int iw[W];
    ldraw<W>(p.g + i, gr);
    ldraw<W>(p.x + i, xr);
    ldiw<W>(p.iw + i, iw);
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      ldraw<W>(p.wy + j * p.ld + i, t);
#pragma unroll
      for (int k = 0; k < W; ++k) a[j][k] = t[k];
      ldraw<W>(p.ws + j * p.ld + i, t);
#pragma unroll
      for (int k = 0; k < W; ++k) b[j][k] = t[k];
    }
    row_terms<W>(a, b, xr, gr, iw, p.tsum, p.theta, p.cf, rv, yf, sa);
#pragma unroll
    for (int j = 0; j < MC; ++j)
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += a[j][k] * rv[k];
        acc[MC + j] += b[j][k] * rv[k];
        acc[2 * MC + j] += yf[k] * a[j][k];
        acc[3 * MC + j] += sa[k] * b[j][k];
        acc[4 * MC + j] += sa[k] * a[j][k];
        acc[5 * MC + j] += b[j][k] * yf[k];
      }
  }
  block_store<NA>(acc, p.part);
}

// ------------------------------------------- v1 / v3: raw fp32 operands, W rows per lane
template <int W>
__device__ __forceinline__ void load_trip(const Args &p, int64_t i, float (&a)[MC][W], float (&b)[MC][W], float (&xr)[W],
                                          float (&gr)[W], int (&iw)[W]) {
  ldraw<W>(p.g + i, gr);
  ldraw<W>(p.x + i, xr);
  ldiw<W>(p.iw + i, iw);
#pragma unroll
  for (int j = 0; j < MC; ++j) {
    ldraw<W>(p.wy + j * p.ld + i, a[j]);
    ldraw<W>(p.ws + j * p.ld + i, b[j]);
  }
}
template <int W>
__device__ __forceinline__ void sum_trip(const Args &p, const float (&a)[MC][W], const float (&b)[MC][W],
                                         const float (&xr)[W], const float (&gr)[W], const int (&iw)[W],
                                         double (&acc)[NA]) {
  double rv[W], yf[W], sa[W];
  row_terms<W>(a, b, xr, gr, iw, p.tsum, p.theta, p.cf, rv, yf, sa);
#pragma unroll
  for (int j = 0; j < MC; ++j)
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const double aj = widen_late(a[j][k]), bj = widen_late(b[j][k]);
      acc[j] += aj * rv[k];
      acc[MC + j] += bj * rv[k];
      acc[2 * MC + j] += yf[k] * aj;
      acc[3 * MC + j] += sa[k] * bj;
      acc[4 * MC + j] += sa[k] * aj;
      acc[5 * MC + j] += bj * yf[k];
    }
}
template <int W>
__global__ __launch_bounds__(256) void v1(Args p) {
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const int64_t nv = p.n / W, stride = (int64_t)gridDim.x * 256;
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    float a[MC][W], b[MC][W], xr[W], gr[W];
    int iw[W];
    load_trip<W>(p, iv * W, a, b, xr, gr, iw);
    sum_trip<W>(p, a, b, xr, gr, iw, acc);
  }
  block_store<NA>(acc, p.part);
}
// ------------------------------------------- v2: ping-pong operand registers
__global__ __launch_bounds__(256) void v2(Args p) {
  constexpr int W = 2;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const int64_t nv = p.n / W, stride = (int64_t)gridDim.x * 256;
  int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float a0[MC][W], b0[MC][W], x0[W], g0[W], a1[MC][W], b1[MC][W], x1[W], g1[W];
  int i0[W], i1[W];
  if (iv < nv) load_trip<W>(p, iv * W, a0, b0, x0, g0, i0);
  while (iv < nv) {
    const int64_t n1 = iv + stride, n2 = n1 + stride;
    // a trip past the end re-reads the last valid one (no branch around the loads)
    load_trip<W>(p, (n1 < nv ? n1 : iv) * W, a1, b1, x1, g1, i1);
    sum_trip<W>(p, a0, b0, x0, g0, i0, acc);
    if (n1 >= nv) break;
    load_trip<W>(p, (n2 < nv ? n2 : n1) * W, a0, b0, x0, g0, i0);
    sum_trip<W>(p, a1, b1, x1, g1, i1, acc);
    iv = n2;
  }
  block_store<NA>(acc, p.part);
}

// ------------------------------------------- v5 / v4: lanes share the accumulators (DPP)
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)(unsigned)hi << 32) | (unsigned)lo);
}
// G lanes of a group (G = 2: lanes l, l^1; G = 4: a quad) split the 2*MC columns: member q sums
// columns [q*H, q*H+H) of Wy and of Ws, H = MC / G, for the rows of all G members.
template <int G>
__global__ __launch_bounds__(256) void vshare(Args p) {
  constexpr int W = 2, H = MC / G, NAL = 6 * H;
  double acc[NAL];
#pragma unroll
  for (int k = 0; k < NAL; ++k) acc[k] = 0.0;
  const int lane = threadIdx.x & 63, q = lane & (G - 1);
  const int64_t nv = p.n / W, stride = (int64_t)gridDim.x * 256;
  // (n is a multiple of 4 * 256 * W in this benchmark: every lane of a group has a trip)
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    float a[MC][W], b[MC][W], xr[W], gr[W];
    int iw[W];
    load_trip<W>(p, iv * W, a, b, xr, gr, iw);
    double rv[W], yf[W], sa[W];
    row_terms<W>(a, b, xr, gr, iw, p.tsum, p.theta, p.cf, rv, yf, sa);
    auto add = [&](const float (&aa)[H][W], const float (&bb)[H][W], const double (&r_)[W], const double (&y_)[W],
                   const double (&s_)[W]) {
#pragma unroll
      for (int jj = 0; jj < H; ++jj)
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const double aj = widen_late(aa[jj][k]), bj = widen_late(bb[jj][k]);
          acc[jj] += aj * r_[k];
          acc[H + jj] += bj * r_[k];
          acc[2 * H + jj] += y_[k] * aj;
          acc[3 * H + jj] += s_[k] * bj;
          acc[4 * H + jj] += s_[k] * aj;
          acc[5 * H + jj] += bj * y_[k];
        }
    };
    // what this lane presents to the member at xor-distance d: its operands for THAT member's columns
    auto pick = [&](int d, float (&oa)[H][W], float (&ob)[H][W]) {
      const int qr = q ^ d;
#pragma unroll
      for (int jj = 0; jj < H; ++jj)
#pragma unroll
        for (int k = 0; k < W; ++k) {
          float va = a[jj][k], vb = b[jj][k];
#pragma unroll
          for (int t = 1; t < G; ++t) {
            va = qr == t ? a[t * H + jj][k] : va;
            vb = qr == t ? b[t * H + jj][k] : vb;
          }
          oa[jj][k] = va, ob[jj][k] = vb;
        }
    };
    float oa[H][W], ob[H][W];
    pick(0, oa, ob);
    add(oa, ob, rv, yf, sa);
#define SHARE_STEP(D, CTRL)                                                      \
    {                                                                            \
      pick(D, oa, ob);                                                           \
      float ra[H][W], rb[H][W];                                                  \
      double r_[W], y_[W], s_[W];                                                \
      _Pragma("unroll") for (int jj = 0; jj < H; ++jj)                           \
      _Pragma("unroll") for (int k = 0; k < W; ++k) {                            \
        ra[jj][k] = dppf<CTRL>(oa[jj][k]);                                       \
        rb[jj][k] = dppf<CTRL>(ob[jj][k]);                                       \
      }                                                                          \
      _Pragma("unroll") for (int k = 0; k < W; ++k) {                            \
        r_[k] = dppd<CTRL>(rv[k]);                                               \
        y_[k] = dppd<CTRL>(yf[k]);                                               \
        s_[k] = dppd<CTRL>(sa[k]);                                               \
      }                                                                          \
      add(ra, rb, r_, y_, s_);                                                   \
    }
    SHARE_STEP(1, 0xB1)
    if constexpr (G == 4) {
      SHARE_STEP(2, 0x4E)
      SHARE_STEP(3, 0x1B)
    }
#undef SHARE_STEP
  }
  // lanes with equal q hold the same slots: reduce over them, then across the waves
  __shared__ double sm[4][G][NAL];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NAL; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) v += __shfl_xor(v, o);
    if (lane < G) sm[w][lane][k] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < G * NAL; e += blockDim.x) {
    const int qq = e / NAL, k = e % NAL;
    const double s = ((sm[0][qq][k] + sm[1][qq][k]) + sm[2][qq][k]) + sm[3][qq][k];
    const int grp = k / H, jj = k % H;
    p.part[(size_t)(grp * MC + qq * H + jj) * MAXB + blockIdx.x] = s;
  }
}

// ---------------------------------------------------------------- host
static hipEvent_t e0, e1;
template <typename F>
float timeit(F &&launch, int reps = 6) {
  for (int r = 0; r < 2; ++r) launch();
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int64_t n = (int64_t)48828 * 2048;   // ~1e8, a multiple of 4*256*2
  const int64_t ld = n;
  float *x, *g, *ws, *wy;
  int8_t *iw;
  double *part;
  CK(hipMalloc(&x, n * 4));
  CK(hipMalloc(&g, n * 4));
  CK(hipMalloc(&ws, (size_t)n * MC * 4));
  CK(hipMalloc(&wy, (size_t)n * MC * 4));
  CK(hipMalloc(&iw, n));
  CK(hipMalloc(&part, (size_t)NA * MAXB * 8));
  {
    // cheap deterministic fill on the host, in pieces
    std::vector<float> h((size_t)1 << 24);
    uint32_t s = 12345u;
    auto fill = [&](float *d, size_t cnt) {
      for (size_t o = 0; o < cnt; o += h.size()) {
        const size_t c = std::min(h.size(), cnt - o);
        for (size_t k = 0; k < c; ++k) {
          s = s * 1664525u + 1013904223u;
          h[k] = (float)((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22));
        }
        CK(hipMemcpy(d + o, h.data(), c * 4, hipMemcpyHostToDevice));
      }
    };
    fill(x, n), fill(g, n), fill(ws, (size_t)n * MC), fill(wy, (size_t)n * MC);
    std::vector<int8_t> hi(n);
    for (int64_t k = 0; k < n; ++k) {
      s = s * 1664525u + 1013904223u;
      hi[k] = (int8_t)((s >> 20) % 3);   // 0, 1, 2: ~1/3 free
    }
    CK(hipMemcpy(iw, hi.data(), n, hipMemcpyHostToDevice));
  }
  Args p;
  p.n = n, p.x = x, p.g = g, p.ws = ws, p.wy = wy, p.iw = iw, p.ld = ld, p.tsum = 0.37, p.theta = 1.7, p.part = part;
  for (int j = 0; j < 2 * MC; ++j) p.cf.a[j] = 0.01 * (j + 1) - 0.2;
  const double gb = ((2.0 * MC + 2) * 4 + 1) * n / 1e9;
  std::vector<double> ref, cur((size_t)NA * MAXB);
  auto sums = [&](int grid) {
    CK(hipMemcpy(cur.data(), part, cur.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> out(NA, 0.0);
    for (int k = 0; k < NA; ++k)
      for (int b = 0; b < grid; ++b) out[k] += cur[(size_t)k * MAXB + b];
    return out;
  };
  auto report = [&](const char *name, int grid, float ms) {
    const std::vector<double> s = sums(grid);
    double err = 0.0;
    if (ref.empty()) ref = s;
    for (int k = 0; k < NA; ++k) err = std::max(err, std::fabs(s[k] - ref[k]) / (std::fabs(ref[k]) + 1e-300));
    printf("%-34s grid %5d  %7.3f ms  %7.1f GB/s  (%.0f%% of 8 TB/s)  max rel diff vs v0 %.1e\n", name, grid, ms,
           gb / ms * 1e3, gb / ms * 1e3 / 80.0, err);
    fflush(stdout);
  };
  for (int pass = 0; pass < 2; ++pass) {
    for (int grid : {512, 768, 1024, 2048}) {
      report("v0 shipped (fp64 operands, W=2)", grid, timeit([&] { hipLaunchKernelGGL(v0, dim3(grid), dim3(256), 0, 0, p); }));
      report("v1 raw fp32 operands, W=2", grid, timeit([&] { hipLaunchKernelGGL(v1<2>, dim3(grid), dim3(256), 0, 0, p); }));
      report("v2 raw + ping-pong prefetch, W=2", grid, timeit([&] { hipLaunchKernelGGL(v2, dim3(grid), dim3(256), 0, 0, p); }));
      report("v3 raw fp32 operands, W=4", grid, timeit([&] { hipLaunchKernelGGL(v1<4>, dim3(grid), dim3(256), 0, 0, p); }));
      report("v5 pair-shared accumulators (DPP)", grid, timeit([&] { hipLaunchKernelGGL(vshare<2>, dim3(grid), dim3(256), 0, 0, p); }));
      report("v4 quad-shared accumulators (DPP)", grid, timeit([&] { hipLaunchKernelGGL(vshare<4>, dim3(grid), dim3(256), 0, 0, p); }));
    }
    printf("\n");
  }
  return 0;
}
