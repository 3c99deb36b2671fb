// Micro-benchmark: does the cache policy of the STORES change what the storing pass of the
// iteration reaches?  24 read streams (nontemporal loads) + NW write streams, n = 1e8 rows, fp64,
// 16 B per lane, grid-stride over 768 workgroups, stores as buffer stores with the gfx950 cache
// policy bits: plain | nt | sc1 (write-through, line dropped from L2) | sc0 sc1 | sc1 nt; and the
// same with 4 rows per lane (two 16-byte accesses per stream and lane: 2 KB per wave and stream).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 store_flavours.hip -o store_flavours
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d2 ldnt(const double *p) { return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p)); }

template <int AUX>
__device__ __forceinline__ void st_pol(double *wave_base, int lane_off_bytes, d2 v) {
  // wave_base: uniform per wave (first row of the wave's chunk); 64 lanes x 32 B at most
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(wave_base, 0, 1 << 20, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), r, lane_off_bytes, 0, AUX);
}

template <int NR, int NW, int AUX, int RP>   // RP = row pairs per lane (1 or 2)
__global__ __launch_bounds__(256) void k(int64_t n, const double *__restrict__ w, double *out, int64_t ld, double *sink) {
  const int64_t nv = n / (2 * RP), stride = (int64_t)gridDim.x * 256;
  const int lane = threadIdx.x & 63;
  d2 acc = {0.0, 0.0};
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    d2 v[NR][RP];
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int q = 0; q < RP; ++q) v[j][q] = ldnt(w + (int64_t)j * ld + iv * 2 * RP + 2 * q);
    d2 s[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) {
      s[q] = d2{1.0, 2.0};
#pragma unroll
      for (int j = 0; j < NR; ++j) s[q] += v[j][q];
      acc += s[q];
    }
    const int64_t wave_iv = __builtin_amdgcn_readfirstlane((int)((iv - lane) & 0x7fffffff)) |
                            ((int64_t)__builtin_amdgcn_readfirstlane((int)((iv - lane) >> 31)) << 31);
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int q = 0; q < RP; ++q)
        st_pol<AUX>(out + (int64_t)j * ld + wave_iv * 2 * RP, lane * 16 * RP + 16 * q, s[q] + (double)j);
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

static hipEvent_t e0, e1;
template <typename F>
float timeit(F &&launch, int reps = 8) {
  for (int r = 0; r < 2; ++r) launch();
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}
template <int NR, int NW, int AUX, int RP>
void run(const char *name, int64_t n, const double *w, double *out, double *sink, int grid) {
  const double gb = (NR + NW) * 8.0 * n / 1e9;
  const float ms = timeit([&] { hipLaunchKernelGGL((k<NR, NW, AUX, RP>), dim3(grid), dim3(256), 0, 0, n, w, out, n, sink); });
  printf("reads %2d writes %d rows/lane %d stores %-8s grid %5d  %7.3f ms %7.1f GB/s\n", NR, NW, 2 * RP, name, grid, ms, gb / ms * 1e3);
  fflush(stdout);
}
template <int NW, int RP>
void suite(int64_t n, const double *w, double *out, double *sink) {
  for (int grid : {768, 1024}) {
    run<24, NW, 0, RP>("plain", n, w, out, sink, grid);
    run<24, NW, 2, RP>("nt", n, w, out, sink, grid);
    run<24, NW, 16, RP>("sc1", n, w, out, sink, grid);
    run<24, NW, 17, RP>("sc0 sc1", n, w, out, sink, grid);
    run<24, NW, 18, RP>("sc1 nt", n, w, out, sink, grid);
  }
}
int main() {
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int64_t n = 100000000;
  double *w, *out, *sink;
  CK(hipMalloc(&w, (size_t)n * 24 * 8));
  CK(hipMalloc(&out, (size_t)n * 8 * 8));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(w, 0, (size_t)n * 24 * 8));
  CK(hipMemset(out, 0, (size_t)n * 8 * 8));
  for (int pass = 0; pass < 2; ++pass) {
    suite<5, 1>(n, w, out, sink);
    suite<7, 1>(n, w, out, sink);
    suite<5, 2>(n, w, out, sink);
    printf("\n");
  }
  return 0;
}
