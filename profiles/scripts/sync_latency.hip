// What does one host round trip cost?  (a) the library's fetch(): tiny kernel -> hipMemcpyAsync D2H
// of 53 doubles -> hipStreamSynchronize; (b) the same kernel writing its results and then a
// sequence flag into host-mapped pinned memory (system-scope release), the host spinning on the
// flag.  Both preceded by a ~0.4 ms streaming kernel, like a pass of the 8-GPU shape, and followed
// by the next launch (the round trip that matters is: results of pass k -> host -> launch k+1).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void stream_k(const double *a, double *part, long n) {
  double s = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += a[i];
  if (s == 123.456) part[0] = s;
}
__global__ void fin_dev(const double *part, double *res, int k) {
  if (threadIdx.x < k) res[threadIdx.x] = part[threadIdx.x] + 1.0;
}
__global__ void fin_host(const double *part, double *hres, unsigned long long *flag, unsigned long long seq, int k) {
  if (threadIdx.x < k) hres[threadIdx.x] = part[threadIdx.x] + 1.0;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
  const long n = 50000000;  // 0.4 GB: ~0.07 ms... use a few of them
  double *a, *part, *res, *hres, *dh;
  unsigned long long *hflag, *dflag;
  CK(hipMalloc(&a, n * 8));
  CK(hipMemset(a, 0, n * 8));
  CK(hipMalloc(&part, 4096 * 8));
  CK(hipMemset(part, 0, 4096 * 8));
  CK(hipMalloc(&res, 64 * 8));
  double *hpin;
  CK(hipHostMalloc(&hpin, 64 * 8));
  CK(hipHostMalloc(&hres, 64 * 8, hipHostMallocMapped));
  CK(hipHostMalloc(&hflag, 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&dh, hres, 0));
  CK(hipHostGetDevicePointer((void **)&dflag, hflag, 0));
  *hflag = 0;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int reps = 2000, k = 53;
  for (int mode = 0; mode < 2; ++mode) {
    for (int big = 0; big < 2; ++big) {
      CK(hipStreamSynchronize(st));
      double t0 = now();
      unsigned long long seq = *hflag;
      for (int r = 0; r < reps; ++r) {
        if (big) hipLaunchKernelGGL(stream_k, dim3(768), dim3(256), 0, st, a, part, n);
        if (mode == 0) {
          hipLaunchKernelGGL(fin_dev, dim3(1), dim3(64), 0, st, part, res, k);
          CK(hipMemcpyAsync(hpin, res, k * 8, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
        } else {
          ++seq;
          hipLaunchKernelGGL(fin_host, dim3(1), dim3(64), 0, st, part, dh, dflag, seq, k);
          while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != seq) {
          }
        }
      }
      double dt = (now() - t0) / reps * 1e6;
      printf("%-44s %s: %8.1f us per round trip\n", mode == 0 ? "kernel + D2H copy + hipStreamSynchronize" : "kernel -> host-mapped results + flag, host spins",
             big ? "behind a 0.4 GB streaming kernel" : "alone                           ", dt);
    }
  }
  CK(hipStreamSynchronize(st));
  return 0;
}
