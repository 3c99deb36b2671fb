// Micro-benchmark: A/B variants of the W'v streaming kernel in ONE process (n = 1e8, col = 10).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off wtv_variants.hip -o wtv_variants
// Variants differ in load flavour (plain / nontemporal), rows in flight per lane, grid size and
// workgroup size.  Prints GB/s of algorithmic bytes (2*col+1)*n*8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int MC = 10;

template <bool NT>
__device__ __forceinline__ double2 ld2(const double *p) {
  if constexpr (NT) {
    const double x = __builtin_nontemporal_load(p);
    const double y = __builtin_nontemporal_load(p + 1);
    return make_double2(x, y);
  } else {
    return *reinterpret_cast<const double2 *>(p);
  }
}
template <bool NT>
__device__ __forceinline__ double2 ld2v(const double *p) {
  if constexpr (NT) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
    return make_double2(v.x, v.y);
  } else {
    return *reinterpret_cast<const double2 *>(p);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// UN = row groups (of 2 rows) in flight per lane and trip
template <bool NT, int UN, int BS>
__global__ __launch_bounds__(BS) void wtv(int64_t n, const double *__restrict__ ws,
                                          const double *__restrict__ wy, int64_t ld,
                                          const double *__restrict__ v, double *part) {
  double acc[2 * MC];
#pragma unroll
  for (int k = 0; k < 2 * MC; ++k) acc[k] = 0.0;
  const int64_t nv = n / 2;
  const int64_t stride = (int64_t)gridDim.x * BS;
  for (int64_t iv = (int64_t)blockIdx.x * BS + threadIdx.x; iv < nv; iv += stride * UN) {
    double2 vv[UN], a[UN][MC], b[UN][MC];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = (iv + u * stride < nv ? iv + u * stride : iv) * 2;
      vv[u] = ld2v<NT>(v + i);
#pragma unroll
      for (int j = 0; j < MC; ++j) {
        a[u][j] = ld2v<NT>(wy + j * ld + i);
        b[u][j] = ld2v<NT>(ws + j * ld + i);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double m = iv + u * stride < nv ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < MC; ++j) {
        acc[j] += m * (a[u][j].x * vv[u].x + a[u][j].y * vv[u].y);
        acc[MC + j] += m * (b[u][j].x * vv[u].x + b[u][j].y * vv[u].y);
      }
    }
  }
  __shared__ double sm[BS / 64][2 * MC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 2 * MC; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) sm[w][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 2 * MC) {
    double s = 0;
    for (int q = 0; q < BS / 64; ++q) s += sm[q][threadIdx.x];
    part[(size_t)threadIdx.x * 16384 + blockIdx.x] = s;
  }
}

template <bool NT, int UN, int BS>
double run(const char *name, int grid, int64_t n, const double *ws, const double *wy, int64_t ld,
           const double *v, double *part) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((wtv<NT, UN, BS>), dim3(grid), dim3(BS), 0, 0, n, ws, wy, ld, v, part);
  CK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((wtv<NT, UN, BS>), dim3(grid), dim3(BS), 0, 0, n, ws, wy, ld, v, part);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double gbs = (2.0 * MC + 1) * n * 8 / (ms * 1e-3) / 1e9;
  printf("%-34s grid %6d  %.3f ms  %.0f GB/s  (%.1f%% of 8 TB/s)\n", name, grid, ms, gbs, gbs / 80.0);
  return gbs;
}

int main() {
  const int64_t n = 100000000, ld = n;
  double *ws, *wy, *v, *part;
  CK(hipMalloc(&ws, (size_t)ld * MC * 8));
  CK(hipMalloc(&wy, (size_t)ld * MC * 8));
  CK(hipMalloc(&v, (size_t)n * 8));
  CK(hipMalloc(&part, (size_t)16384 * 2 * MC * 8));
  CK(hipMemset(ws, 0, (size_t)ld * MC * 8));
  CK(hipMemset(wy, 0, (size_t)ld * MC * 8));
  CK(hipMemset(v, 0, (size_t)n * 8));
  for (int pass = 0; pass < 2; ++pass) {
    run<false, 1, 256>("plain  un1 bs256", 2048, n, ws, wy, ld, v, part);
    run<false, 1, 256>("plain  un1 bs256", 1024, n, ws, wy, ld, v, part);
    run<false, 1, 256>("plain  un1 bs256", 4096, n, ws, wy, ld, v, part);
    run<false, 1, 256>("plain  un1 bs256", 8192, n, ws, wy, ld, v, part);
    run<true, 1, 256>("nt     un1 bs256", 2048, n, ws, wy, ld, v, part);
    run<true, 1, 256>("nt     un1 bs256", 4096, n, ws, wy, ld, v, part);
    run<false, 2, 256>("plain  un2 bs256", 2048, n, ws, wy, ld, v, part);
    run<true, 2, 256>("nt     un2 bs256", 2048, n, ws, wy, ld, v, part);
    run<false, 1, 512>("plain  un1 bs512", 1024, n, ws, wy, ld, v, part);
    run<false, 1, 512>("plain  un1 bs512", 2048, n, ws, wy, ld, v, part);
    run<true, 1, 512>("nt     un1 bs512", 2048, n, ws, wy, ld, v, part);
    run<false, 1, 128>("plain  un1 bs128", 4096, n, ws, wy, ld, v, part);
    run<false, 1, 64>("plain  un1 bs64", 8192, n, ws, wy, ld, v, part);
    printf("\n");
  }
  return 0;
}
