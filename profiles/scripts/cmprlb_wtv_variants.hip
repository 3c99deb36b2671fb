// Micro-benchmark: A/B variants of the fused cmprlb + W'r (+ formk new-row sums) streaming kernel
// in ONE process (n = 1e8, col = 10, fp64).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off cmprlb_wtv_variants.hip -o cwv
// Variants: NEWROW on/off, r store on/off, vector loads (x,z,iwhere) on/off, rows per lane,
// workgroup size, grid size.  Prints GB/s of the algorithmic bytes of the full kernel
// ((2*col+4)*8 + 4) per row, so that "less work" variants show what each stream costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int MC = 10;
struct Coef {
  double a[2 * MC];
};

template <int W>
__device__ __forceinline__ void ldn(const double *p, double (&o)[W]) {
  if constexpr (W == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
    o[0] = v.x, o[1] = v.y;
  } else {
    o[0] = __builtin_nontemporal_load(p);
  }
}
template <int W>
__device__ __forceinline__ void ldin(const int *p, int (&o)[W]) {
  if constexpr (W == 2) {
    int2 v = *reinterpret_cast<const int2 *>(p);
    o[0] = v.x, o[1] = v.y;
  } else {
    o[0] = *p;
  }
}
template <int W>
__device__ __forceinline__ void stn(double *p, const double (&o)[W]) {
  if constexpr (W == 2)
    *reinterpret_cast<double2 *>(p) = make_double2(o[0], o[1]);
  else
    *p = o[0];
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// flags: NEWROW, STORE (write r), VECS (load x, z, iwhere)
template <int W, int BS, bool NEWROW, bool STORE, bool VECS>
__global__ __launch_bounds__(BS) void cwv(int64_t n, const double *__restrict__ x,
                                          const double *__restrict__ g,
                                          const double *__restrict__ z, double *r,
                                          const int *__restrict__ iwhere,
                                          const double *__restrict__ ws,
                                          const double *__restrict__ wy, int64_t ld, int col,
                                          double theta, Coef cf, double *part) {
  constexpr int NA = NEWROW ? 6 * MC : 2 * MC;
  double acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  const int64_t nv = n / W;
  const int64_t stride = (int64_t)gridDim.x * BS;
  for (int64_t iv = (int64_t)blockIdx.x * BS + threadIdx.x; iv < nv; iv += stride) {
    const int64_t i = iv * W;
    double xv[W], gv[W], zv[W], rv[W], a[MC][W], b[MC][W];
    int iw[W];
    ldn<W>(g + i, gv);
    if constexpr (VECS) {
      ldn<W>(x + i, xv);
      ldn<W>(z + i, zv);
      ldin<W>(iwhere + i, iw);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) xv[k] = 1.0, zv[k] = 2.0, iw[k] = (int)(i & 1) - 1;
    }
#pragma unroll
    for (int j = 0; j < MC; ++j) {
      ldn<W>(wy + j * ld + i, a[j]);
      ldn<W>(ws + j * ld + i, b[j]);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      double rr = -theta * (zv[k] - xv[k]) - gv[k];
#pragma unroll
      for (int j = 0; j < MC; ++j)
        if (j < col) rr = rr + a[j][k] * cf.a[j] + b[j][k] * cf.a[MC + j];
      rv[k] = iw[k] <= 0 ? rr : 0.0;
    }
    if constexpr (STORE) stn<W>(r + i, rv);
#pragma unroll
    for (int j = 0; j < MC; ++j)
#pragma unroll
      for (int k = 0; k < W; ++k) {
        acc[j] += a[j][k] * rv[k];
        acc[MC + j] += b[j][k] * rv[k];
      }
    if constexpr (NEWROW) {
      double yf[W], sa[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        double yn = 0.0, sn = 0.0;
#pragma unroll
        for (int j = 0; j < MC; ++j)
          if (j == col - 1) yn = a[j][k], sn = b[j][k];
        yf[k] = iw[k] <= 0 ? yn : 0.0;
        sa[k] = iw[k] <= 0 ? 0.0 : sn;
      }
#pragma unroll
      for (int j = 0; j < MC; ++j)
#pragma unroll
        for (int k = 0; k < W; ++k) {
          acc[2 * MC + j] += yf[k] * a[j][k];
          acc[3 * MC + j] += sa[k] * b[j][k];
          acc[4 * MC + j] += sa[k] * a[j][k];
          acc[5 * MC + j] += b[j][k] * yf[k];
        }
    }
  }
  __shared__ double sm[BS / 64][NA];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) sm[w][k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < NA; k += BS) {
    double s = 0.0;
    for (int q = 0; q < BS / 64; ++q) s += sm[q][k];
    part[(size_t)k * 16384 + blockIdx.x] = s;
  }
}

struct Bufs {
  double *x, *g, *z, *r, *ws, *wy, *part;
  int *iw;
  int64_t n, ld;
};

template <int W, int BS, bool NEWROW, bool STORE, bool VECS>
void run(const char *name, int grid, const Bufs &B) {
  Coef cf;
  for (int k = 0; k < 2 * MC; ++k) cf.a[k] = 0.5;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < 2; ++k)
    hipLaunchKernelGGL((cwv<W, BS, NEWROW, STORE, VECS>), dim3(grid), dim3(BS), 0, 0, B.n, B.x, B.g,
                       B.z, B.r, B.iw, B.ws, B.wy, B.ld, MC, 1.0, cf, B.part);
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int k = 0; k < reps; ++k)
    hipLaunchKernelGGL((cwv<W, BS, NEWROW, STORE, VECS>), dim3(grid), dim3(BS), 0, 0, B.n, B.x, B.g,
                       B.z, B.r, B.iw, B.ws, B.wy, B.ld, MC, 1.0, cf, B.part);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double full = ((2.0 * MC + 4) * 8 + 4) * B.n;
  const double own = ((2.0 * MC + 1 + (STORE ? 1 : 0) + (VECS ? 2 : 0)) * 8 + (VECS ? 4 : 0)) * B.n;
  printf("%-44s grid %5d  %7.3f ms  %7.1f GB/s(full)  %7.1f GB/s(own bytes)\n", name, grid, ms,
         full / ms / 1e6, own / ms / 1e6);
  fflush(stdout);
}

int main() {
  Bufs B;
  B.n = 100000000, B.ld = B.n;
  CK(hipMalloc(&B.ws, (size_t)B.ld * MC * 8));
  CK(hipMalloc(&B.wy, (size_t)B.ld * MC * 8));
  CK(hipMalloc(&B.x, (size_t)B.n * 8));
  CK(hipMalloc(&B.g, (size_t)B.n * 8));
  CK(hipMalloc(&B.z, (size_t)B.n * 8));
  CK(hipMalloc(&B.r, (size_t)B.n * 8));
  CK(hipMalloc(&B.iw, (size_t)B.n * 4));
  CK(hipMalloc(&B.part, (size_t)16384 * 6 * MC * 8));
  CK(hipMemset(B.ws, 0, (size_t)B.ld * MC * 8));
  CK(hipMemset(B.wy, 0, (size_t)B.ld * MC * 8));
  CK(hipMemset(B.x, 0, (size_t)B.n * 8));
  CK(hipMemset(B.g, 0, (size_t)B.n * 8));
  CK(hipMemset(B.z, 0, (size_t)B.n * 8));
  CK(hipMemset(B.r, 0, (size_t)B.n * 8));
  CK(hipMemset(B.iw, 0, (size_t)B.n * 4));
  for (int pass = 0; pass < 2; ++pass) {
    run<2, 256, true, true, true>("newrow store vecs w2 bs256 (shipped)", 2048, B);
    run<2, 256, true, true, true>("newrow store vecs w2 bs256", 512, B);
    run<2, 256, true, true, true>("newrow store vecs w2 bs256", 1024, B);
    run<2, 256, true, true, true>("newrow store vecs w2 bs256", 4096, B);
    run<2, 256, true, true, true>("newrow store vecs w2 bs256", 8192, B);
    run<2, 512, true, true, true>("newrow store vecs w2 bs512", 1024, B);
    run<2, 512, true, true, true>("newrow store vecs w2 bs512", 2048, B);
    run<2, 128, true, true, true>("newrow store vecs w2 bs128", 4096, B);
    run<1, 256, true, true, true>("newrow store vecs w1 bs256", 2048, B);
    run<1, 256, true, true, true>("newrow store vecs w1 bs256", 4096, B);
    run<1, 512, true, true, true>("newrow store vecs w1 bs512", 2048, B);
    run<2, 256, false, true, true>("       store vecs w2 bs256", 2048, B);
    run<2, 256, false, true, true>("       store vecs w2 bs256", 3072, B);
    run<2, 256, true, false, true>("newrow       vecs w2 bs256", 2048, B);
    run<2, 256, true, true, false>("newrow store      w2 bs256", 2048, B);
    run<2, 256, false, false, false>("                  w2 bs256 (= W'v)", 2048, B);
    run<2, 256, false, false, true>("             vecs w2 bs256", 2048, B);
    run<2, 256, false, true, false>("       store      w2 bs256", 2048, B);
    printf("\n");
  }
  return 0;
}
