// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the compact-W passes use
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B per
// lane) ... Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel reads a KNOWN number of bytes from a buffer much larger than the Infinity Cache:
//   rd16_nt / rd16_plain   16 B per lane, contiguous (the natural-order passes)
//   rd8_nt / rd8_plain      8 B per lane, contiguous (the row vectors of the compact passes)
//   half8_nt / half8_plain  8 B per lane, lanes 0..31 contiguous, lanes 32..63 all read the first element of the
//                           128-row tile (the W entries of the compact passes at half of the rows free): known
//                           bytes = 256 B per wave instruction (+ nothing for the broadcast)
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o bin/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ d2 l16(const double *p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
  else return *reinterpret_cast<const d2 *>(p);
}
template <bool NT> __device__ __forceinline__ double l8(const double *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT> __global__ __launch_bounds__(256) void rd16(const double *__restrict__ a, int64_t n, double *sink) {
  double s = 0;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * 512) { d2 v = l16<NT>(a + i); s += v.x + v.y; }
  if (s == 1.2345) *sink = s;
}
template <bool NT> __global__ __launch_bounds__(256) void rd8(const double *__restrict__ a, int64_t n, double *sink) {
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += l8<NT>(a + i);
  if (s == 1.2345) *sink = s;
}
// tiles of 128 elements: lanes 0..31 of a wave read elements 0..31 of "their" half-run, lanes 32..63 element 0
template <bool NT> __global__ __launch_bounds__(256) void half8(const double *__restrict__ a, int64_t n, double *sink) {
  double s = 0;
  const int lane = threadIdx.x & 63;
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w * 64 < n; w += (int64_t)gridDim.x * 4) {
    // wave-trip w <-> half (w & 1) of tile (w >> 1): run of 32 elements at tile * 128 + half * 32
    const int64_t base = (w >> 1) * 128;
    const int64_t i = lane < 32 ? base + (w & 1) * 32 + lane : base;
    s += l8<NT>(a + i);
  }
  if (s == 1.2345) *sink = s;
}
int main() {
  const int64_t n = 1ll << 30;  // 8 GiB of doubles
  double *a, *sink;
  CK(hipMalloc(&a, n * 8));
  CK(hipMemset(a, 0, n * 8));
  CK(hipMalloc(&sink, 8));
  const int g = 2048;
  for (int rep = 0; rep < 3; ++rep) {
    rd16<true><<<g, 256>>>(a, n, sink);
    rd16<false><<<g, 256>>>(a, n, sink);
    rd8<true><<<g, 256>>>(a, n, sink);
    rd8<false><<<g, 256>>>(a, n, sink);
    half8<true><<<g, 256>>>(a, n, sink);
    half8<false><<<g, 256>>>(a, n, sink);
  }
  CK(hipDeviceSynchronize());
  printf("known bytes per launch: rd16 %lld  rd8 %lld  half8 %lld (useful; every second 256-byte half of each 1 KiB tile)\n",
         (long long)(n * 8), (long long)(n * 8), (long long)(n / 64 * 256));
  return 0;
}
