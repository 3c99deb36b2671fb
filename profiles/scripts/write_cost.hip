// Micro-benchmark: what does a write stream cost inside a read-dominated streaming kernel on
// MI355X?  NR read streams + NW write streams of n doubles each (n = 1e8), 16 B per lane.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 write_cost.hip -o write_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NR, int NW, bool NTS>
__global__ __launch_bounds__(256) void k(int64_t n, const double *__restrict__ in, double *out,
                                         int64_t ld, double *sink) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  d2 acc = {0.0, 0.0};
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    d2 v[NR > 0 ? NR : 1];
#pragma unroll
    for (int j = 0; j < NR; ++j)
      v[j] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(in + j * ld + iv * 2));
    d2 s = {1.0, 2.0};
#pragma unroll
    for (int j = 0; j < NR; ++j) s += v[j];
    acc += s;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      d2 o = s + (double)j;
      if constexpr (NTS)
        __builtin_nontemporal_store(o, reinterpret_cast<d2 *>(out + j * ld + iv * 2));
      else
        *reinterpret_cast<d2 *>(out + j * ld + iv * 2) = o;
    }
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

template <int NR, int NW, bool NTS>
void run(int64_t n, const double *in, double *out, double *sink, int grid = 2048) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NR, NW, NTS>), dim3(grid), dim3(256), 0, 0, n, in, out, n, sink);
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<NR, NW, NTS>), dim3(grid), dim3(256), 0, 0, n, in, out, n, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("reads %2d writes %d %s grid %4d  %7.3f ms  %7.1f GB/s\n", NR, NW, NTS ? "nt-store" : "plain   ", grid, ms,
         (NR + NW) * 8.0 * n / ms / 1e6);
  fflush(stdout);
}

int main() {
  const int64_t n = 100000000;
  double *in, *out, *sink;
  CK(hipMalloc(&in, (size_t)n * 24 * 8));
  CK(hipMalloc(&out, (size_t)n * 8 * 8));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 0, (size_t)n * 24 * 8));
  CK(hipMemset(out, 0, (size_t)n * 8 * 8));
  for (int pass = 0; pass < 2; ++pass) {
    run<24, 0, false>(n, in, out, sink);
    run<24, 1, false>(n, in, out, sink);
    run<24, 1, true>(n, in, out, sink);
    run<24, 2, false>(n, in, out, sink);
    run<24, 4, false>(n, in, out, sink);
    run<24, 4, true>(n, in, out, sink);
    run<20, 4, false>(n, in, out, sink);
    run<12, 0, false>(n, in, out, sink);
    run<12, 1, false>(n, in, out, sink);
    run<12, 4, false>(n, in, out, sink);
    run<4, 0, false>(n, in, out, sink);
    run<4, 4, false>(n, in, out, sink);
    run<1, 1, false>(n, in, out, sink);
    run<1, 1, true>(n, in, out, sink);
    run<0, 1, false>(n, in, out, sink);
    run<0, 4, false>(n, in, out, sink);
    run<0, 4, true>(n, in, out, sink);
    // fewer, resident workgroups (what the library uses for its fp64 passes over W)
    run<24, 0, false>(n, in, out, sink, 768);
    run<24, 1, true>(n, in, out, sink, 768);
    run<24, 4, true>(n, in, out, sink, 768);
    run<24, 7, true>(n, in, out, sink, 768);
    run<24, 7, true>(n, in, out, sink, 2048);
    run<12, 0, false>(n, in, out, sink, 768);
    printf("\n");
  }
  return 0;
}
