// Micro-benchmark: the ceiling of a kernel that STORES on MI355X.
//
// MI355X_MICROARCH.md quotes "6.29 TB/s measured (float4 copy, 79 %)" without giving the grid.
// Round 1's write_cost.hip measured 5.0-5.3 TB/s for a copy written like the library's passes
// (grid-stride, 16 B per lane, 2048 workgroups).  This sweep looks for the copy shape that
// reaches the guide's number, and for the store policy / shape that lifts a read-dominated
// pass with a few store streams (the library's subsm_update_kernel: 24 reads + 7 writes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 copy_ceiling.hip -o copy_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// store policies: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt
template <int POL>
__device__ __forceinline__ void store16(f4 *p, f4 v) {
  if constexpr (POL == 0) *p = v;
  else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
  else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <bool NT>
__device__ __forceinline__ f4 load16(const f4 *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// (A) the classic copy: one float4 per thread, no loop
template <bool NTL, int POL>
__global__ void copy_flat(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) store16<POL>(out + i, load16<NTL>(in + i));
}
// (B) grid-stride, U independent loads in flight per lane before the stores
template <int U, bool NTL, int POL>
__global__ void copy_stride(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < nv; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load16<NTL>(in + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) store16<POL>(out + i + u * stride, v[u]);
  }
  for (; i < nv; i += stride) store16<POL>(out + i, load16<NTL>(in + i));
}
// (C) each workgroup owns a CONTIGUOUS slab (nv / grid float4s), walks it in U-deep steps
template <int U, bool NTL, int POL>
__global__ void copy_slab(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv) {
  const size_t per = (nv + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < nv ? b0 + per : nv;
  size_t i = b0 + threadIdx.x;
  for (; i + (U - 1) * blockDim.x < b1; i += U * blockDim.x) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load16<NTL>(in + i + u * blockDim.x);
#pragma unroll
    for (int u = 0; u < U; ++u) store16<POL>(out + i + u * blockDim.x, v[u]);
  }
  for (; i < b1; i += blockDim.x) store16<POL>(out + i, load16<NTL>(in + i));
}
__global__ void fill_k(f4 *out, size_t nv) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride)
    out[i] = f4{1.f, 2.f, 3.f, 4.f};
}
template <bool NTL>
__global__ void read_k(const f4 *__restrict__ in, size_t nv, float *sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) a += load16<NTL>(in + i);
  if (a.x + a.y + a.z + a.w == 12345.678f) sink[0] = a.x;
}

// (D) the library's storing pass in miniature: NR read streams + NW write streams of n doubles,
//     16 B per lane; SLAB = each workgroup owns a contiguous range of rows instead of striding;
//     stores issued after all loads of the trip
template <int NR, int NW, int POL, bool SLAB>
__global__ __launch_bounds__(256) void mixed(int64_t n, const double *__restrict__ in, double *out,
                                             int64_t ld, double *sink) {
  const int64_t nv = n / 2;
  int64_t iv, end, step;
  if constexpr (SLAB) {
    const int64_t per = ((nv + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    iv = (int64_t)blockIdx.x * per + threadIdx.x;
    end = (int64_t)(blockIdx.x + 1) * per < nv ? (int64_t)(blockIdx.x + 1) * per : nv;
    step = 256;
  } else {
    iv = (int64_t)blockIdx.x * 256 + threadIdx.x;
    end = nv;
    step = (int64_t)gridDim.x * 256;
  }
  d2 acc = {0.0, 0.0};
  for (; iv < end; iv += step) {
    d2 v[NR];
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
      f4 of = __builtin_bit_cast(f4, o);
      store16<POL>(reinterpret_cast<f4 *>(out + j * ld + iv * 2), of);
    }
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
static const char *POLN[] = {"plain", "nt", "sc1", "sc0sc1", "sc1nt"};

template <bool NTL, int POL>
void copy_suite(const f4 *in, f4 *out, size_t nv, double gb) {
  for (int blk : {256, 512, 1024}) {
    const float ms = timeit([&] { hipLaunchKernelGGL((copy_flat<NTL, POL>), dim3((unsigned)((nv + blk - 1) / blk)), dim3(blk), 0, 0, in, out, nv); });
    printf("copy flat          ld %-5s st %-6s block %4d grid %8zu  %7.3f ms %7.1f GB/s\n", NTL ? "nt" : "plain", POLN[POL], blk, (nv + blk - 1) / blk, ms, gb / ms * 1e3);
  }
  for (int grid : {512, 1024, 2048, 4096, 16384}) {
    float ms = timeit([&] { hipLaunchKernelGGL((copy_stride<1, NTL, POL>), dim3(grid), dim3(256), 0, 0, in, out, nv); });
    printf("copy stride U1     ld %-5s st %-6s block  256 grid %8d  %7.3f ms %7.1f GB/s\n", NTL ? "nt" : "plain", POLN[POL], grid, ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((copy_stride<4, NTL, POL>), dim3(grid), dim3(256), 0, 0, in, out, nv); });
    printf("copy stride U4     ld %-5s st %-6s block  256 grid %8d  %7.3f ms %7.1f GB/s\n", NTL ? "nt" : "plain", POLN[POL], grid, ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((copy_stride<8, NTL, POL>), dim3(grid), dim3(256), 0, 0, in, out, nv); });
    printf("copy stride U8     ld %-5s st %-6s block  256 grid %8d  %7.3f ms %7.1f GB/s\n", NTL ? "nt" : "plain", POLN[POL], grid, ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((copy_slab<4, NTL, POL>), dim3(grid), dim3(256), 0, 0, in, out, nv); });
    printf("copy slab   U4     ld %-5s st %-6s block  256 grid %8d  %7.3f ms %7.1f GB/s\n", NTL ? "nt" : "plain", POLN[POL], grid, ms, gb / ms * 1e3);
  }
  fflush(stdout);
}

template <int NR, int NW, int POL, bool SLAB>
void mixed_run(int64_t n, const double *in, double *out, double *sink, int grid) {
  const float ms = timeit([&] { hipLaunchKernelGGL((mixed<NR, NW, POL, SLAB>), dim3(grid), dim3(256), 0, 0, n, in, out, n, sink); });
  printf("mixed reads %2d writes %d st %-6s %-6s grid %5d  %7.3f ms %7.1f GB/s\n", NR, NW, POLN[POL], SLAB ? "slab" : "stride", grid, ms,
         (NR + NW) * 8.0 * n / ms / 1e6);
  fflush(stdout);
}

int main() {
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  {
    const size_t bytes = (size_t)4 << 30;  // 4 GiB in, 4 GiB out: far beyond the 256 MiB Infinity Cache
    const size_t nv = bytes / 16;
    f4 *in, *out;
    float *sink;
    CK(hipMalloc(&in, bytes));
    CK(hipMalloc(&out, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, bytes));
    CK(hipMemset(out, 0, bytes));
    const double gb = 2.0 * bytes / 1e9;
    float ms = timeit([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); });
    printf("hipMemcpyAsync D2D 4 GiB                                        %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, out, nv); });
    printf("fill (plain stores) grid 2048                                   %7.3f ms %7.1f GB/s\n", ms, bytes / 1e9 / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(read_k<false>, dim3(2048), dim3(256), 0, 0, in, nv, sink); });
    printf("read (plain loads) grid 2048                                    %7.3f ms %7.1f GB/s\n", ms, bytes / 1e9 / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(read_k<true>, dim3(2048), dim3(256), 0, 0, in, nv, sink); });
    printf("read (nt loads) grid 2048                                       %7.3f ms %7.1f GB/s\n", ms, bytes / 1e9 / ms * 1e3);
    copy_suite<false, 0>(in, out, nv, gb);
    copy_suite<true, 1>(in, out, nv, gb);
    copy_suite<true, 0>(in, out, nv, gb);
    copy_suite<false, 1>(in, out, nv, gb);
    copy_suite<true, 2>(in, out, nv, gb);
    copy_suite<true, 3>(in, out, nv, gb);
    copy_suite<true, 4>(in, out, nv, gb);
    CK(hipFree(in));
    CK(hipFree(out));
    CK(hipFree(sink));
  }
  {
    const int64_t n = 100000000;
    double *in, *out, *sink;
    CK(hipMalloc(&in, (size_t)n * 24 * 8));
    CK(hipMalloc(&out, (size_t)n * 8 * 8));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 0, (size_t)n * 24 * 8));
    CK(hipMemset(out, 0, (size_t)n * 8 * 8));
    for (int grid : {512, 768, 1024, 2048}) {
      mixed_run<24, 0, 0, false>(n, in, out, sink, grid);
      mixed_run<24, 7, 0, false>(n, in, out, sink, grid);
      mixed_run<24, 7, 1, false>(n, in, out, sink, grid);
      mixed_run<24, 7, 2, false>(n, in, out, sink, grid);
      mixed_run<24, 7, 3, false>(n, in, out, sink, grid);
      mixed_run<24, 7, 4, false>(n, in, out, sink, grid);
      mixed_run<24, 0, 0, true>(n, in, out, sink, grid);
      mixed_run<24, 7, 0, true>(n, in, out, sink, grid);
      mixed_run<24, 7, 1, true>(n, in, out, sink, grid);
      mixed_run<24, 7, 2, true>(n, in, out, sink, grid);
    }
  }
  return 0;
}
