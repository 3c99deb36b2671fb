// Stand-alone check of the workgroup reduction epilogue every kernel of the library ends with
// (lbfgsb_amd/csrc/device_util.hpp: block_reduce_store, wave_sum / wave_min / wave_max).  Each lane
// contributes small INTEGER-valued doubles, so every sum is exact whatever the order of the additions
// and the result must EQUAL the host's, slot by slot and workgroup by workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tests/reduce_check.hip -o reduce_check
// Prints "reduce_check ok" and exits 0, or the first mismatch and exits 1.  Test infrastructure only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lbfgsb_amd/csrc/device_util.hpp"

namespace {
constexpr int BLOCK = 256, NBLK = 5, PSTRIDE = 8;

__host__ __device__ inline double contrib(int blk, int tid, int k) {
  // integer in [-500, 500], different for every (block, lane, slot)
  const unsigned h = (unsigned)(blk * 7919 + tid * 104729 + k * 1299709 + 12345);
  return (double)((int)((h * 2654435761u) >> 22) % 1001 - 500);
}

// the constants as the library's call sites pass them (literals after inlining)
template <int K, int NSUM, int NMIN, int NMAX>
__global__ __launch_bounds__(BLOCK) void red_kernel_c(double *part) {
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = contrib(blockIdx.x, threadIdx.x, k);
  lbk::block_reduce_store<K>(acc, NSUM, NMIN, NMAX, part, PSTRIDE);
}
__global__ __launch_bounds__(BLOCK) void wave_kernel(double *out) {
  const double v = contrib(blockIdx.x, threadIdx.x, 3);
  const double s = lbk::wave_sum(v), mn = lbk::wave_min(v), mx = lbk::wave_max(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // every lane must hold the result: lane 0 and lane 37 report
  if (lane == 0 || lane == 37) {
    double *o = out + ((blockIdx.x * 4 + w) * 2 + (lane == 37)) * 3;
    o[0] = s, o[1] = mn, o[2] = mx;
  }
}

int fails = 0;
template <int K, int NSUM, int NMIN, int NMAX>
void run(double *d_part, std::vector<double> &h) {
  static_assert(NSUM + NMIN + NMAX <= K, "slots");
  (void)hipMemset(d_part, 0xff, sizeof(double) * K * PSTRIDE);
  hipLaunchKernelGGL((red_kernel_c<K, NSUM, NMIN, NMAX>), dim3(NBLK), dim3(BLOCK), 0, 0, d_part);
  if (hipDeviceSynchronize() != hipSuccess) {
    std::printf("K=%d: kernel failed\n", K);
    fails++;
    return;
  }
  (void)hipMemcpy(h.data(), d_part, sizeof(double) * K * PSTRIDE, hipMemcpyDeviceToHost);
  for (int b = 0; b < NBLK; ++b)
    for (int k = 0; k < NSUM + NMIN + NMAX; ++k) {
      double e = k < NSUM ? 0.0 : (k < NSUM + NMIN ? HUGE_VAL : -HUGE_VAL);
      for (int t = 0; t < BLOCK; ++t) {
        const double c = contrib(b, t, k);
        e = k < NSUM ? e + c : (k < NSUM + NMIN ? std::fmin(e, c) : std::fmax(e, c));
      }
      const double got = h[(size_t)k * PSTRIDE + b];
      if (!(got == e)) {
        if (fails < 10) std::printf("K=%d (%d,%d,%d) block %d slot %d: got %.17g expected %.17g\n", K, NSUM, NMIN, NMAX, b, k, got, e);
        fails++;
      }
    }
}
}  // namespace

int main() {
  double *d_part = nullptr;
  if (hipMalloc(&d_part, sizeof(double) * 256 * PSTRIDE) != hipSuccess) return 2;
  std::vector<double> h(256 * PSTRIDE);
  // shapes the library uses: pure sums, sums + min + max, maxima only, one slot, odd sizes around the
  // switch to the reduce-scatter form (K >= 16) and its padding to a multiple of 16
  run<1, 1, 0, 0>(d_part, h);
  run<1, 0, 1, 0>(d_part, h);
  run<1, 0, 0, 1>(d_part, h);
  run<2, 1, 0, 1>(d_part, h);
  run<3, 2, 1, 0>(d_part, h);
  run<4, 3, 1, 0>(d_part, h);
  run<5, 0, 0, 5>(d_part, h);
  run<11, 9, 1, 1>(d_part, h);
  run<15, 15, 0, 0>(d_part, h);
  run<16, 16, 0, 0>(d_part, h);
  run<17, 15, 1, 1>(d_part, h);
  run<21, 21, 0, 0>(d_part, h);
  run<31, 29, 1, 1>(d_part, h);   // update_scan, MC = 5
  run<51, 49, 1, 1>(d_part, h);   // update_scan, MC = 10
  run<55, 53, 1, 1>(d_part, h);   // update_scan NEWROW, MC = 5
  run<60, 60, 0, 0>(d_part, h);   // cmprlb_wtv NEWROW, MC = 10
  run<91, 89, 1, 1>(d_part, h);   // update_scan, MC = 20
  run<95, 93, 1, 1>(d_part, h);   // update_scan NEWROW, MC = 10
  run<139, 137, 1, 1>(d_part, h); // update_scan, MC = 32
  run<175, 173, 1, 1>(d_part, h); // update_scan NEWROW, MC = 20
  run<192, 192, 0, 0>(d_part, h); // cmprlb_wtv NEWROW, MC = 32
  // whole-wave helpers
  {
    double *d_out = nullptr;
    (void)hipMalloc(&d_out, sizeof(double) * NBLK * 4 * 2 * 3);
    hipLaunchKernelGGL(wave_kernel, dim3(NBLK), dim3(BLOCK), 0, 0, d_out);
    std::vector<double> o(NBLK * 4 * 2 * 3);
    if (hipMemcpy(o.data(), d_out, sizeof(double) * o.size(), hipMemcpyDeviceToHost) != hipSuccess) fails++;
    for (int b = 0; b < NBLK; ++b)
      for (int w = 0; w < 4; ++w) {
        double s = 0, mn = HUGE_VAL, mx = -HUGE_VAL;
        for (int l = 0; l < 64; ++l) {
          const double c = contrib(b, w * 64 + l, 3);
          s += c, mn = std::fmin(mn, c), mx = std::fmax(mx, c);
        }
        for (int q = 0; q < 2; ++q) {
          const double *g = &o[((b * 4 + w) * 2 + q) * 3];
          if (g[0] != s || g[1] != mn || g[2] != mx) {
            if (fails < 10) std::printf("wave helpers block %d wave %d lane %d: %g %g %g vs %g %g %g\n", b, w, q ? 37 : 0, g[0], g[1], g[2], s, mn, mx);
            fails++;
          }
        }
      }
    (void)hipFree(d_out);
  }
  (void)hipFree(d_part);
  if (fails) {
    std::printf("reduce_check FAILED: %d mismatches\n", fails);
    return 1;
  }
  std::printf("reduce_check ok\n");
  return 0;
}
