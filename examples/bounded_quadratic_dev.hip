// examples/bounded_quadratic_dev.hip -- a caller that keeps everything on the GPU: its own objective as a HIP
// kernel on the solver's stream, the device-pointer entry with ping-pong iterate buffers
// (lbfgsb_hip_setulb_dev_pp), the reference's reverse-communication loop around it (test/driver1.f90:263-292).
// Problem: the separable bounded quadratic of BASELINE.md section 3,
//     f(x) = 1/2 sum_i a_i (x_i - c_i)^2,  -1 <= x_i <= 1,  x0 = 0,
//     a_i = 1 + 99 ((7919 i) mod 10007) / 10006,  c_i = -2 + 4 ((104729 i) mod 100003) / 100002   (i = 1..n)
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/bounded_quadratic_dev.hip -Llbfgsb_amd -llbfgsb_hip \
//         -Wl,-rpath,$PWD/lbfgsb_amd -o bounded_quadratic_dev
//   ./bounded_quadratic_dev [n = 1000000] [m = 10] [iterations = 30]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lbfgsb_hip.h"

#define HIP_OK(call)                                                                 \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

// g = a (x - c); per-block partial sums of f (summed on the host: a caller is free in how it reduces f)
__global__ void objective(long n, const double *__restrict__ x, double *__restrict__ g, double *part) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long k = i + 1;
    const double a = 1.0 + 99.0 * (double)((7919 * k) % 10007) / 10006.0;
    const double c = -2.0 + 4.0 * (double)((104729 * k) % 100003) / 100002.0;
    const double d = x[i] - c;
    g[i] = a * d;
    acc += a * d * d;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 1000000;
  const int m = argc > 2 ? std::atoi(argv[2]) : 10;
  const int max_iter = argc > 3 ? std::atoi(argv[3]) : 30;
  const int blocks = 1024;

  lbfgsb_hip_ctx *ctx = nullptr;
  int rc = lbfgsb_hip_create(n, n, 0, m, LBFGSB_F_NO_RETURN_SYNC, /*device*/ 0, /*stream*/ nullptr, &ctx);
  if (rc != LBFGSB_OK) {
    std::fprintf(stderr, "lbfgsb_hip_create: %d (%s)\n", rc, lbfgsb_hip_last_error());
    return 1;
  }
  // LBFGSB_F_NO_RETURN_SYNC: the objective runs on the solver's own stream, so an 'FG' return needs no host sync
  hipStream_t stream = (hipStream_t)lbfgsb_hip_get_stream(ctx);

  double *x[2], *g[2], *l, *u, *part;
  int32_t *nbd;
  const size_t vb = (size_t)n * sizeof(double);
  for (int k = 0; k < 2; ++k) {
    HIP_OK(hipMalloc(&x[k], vb));
    HIP_OK(hipMalloc(&g[k], vb));
  }
  HIP_OK(hipMalloc(&l, vb));
  HIP_OK(hipMalloc(&u, vb));
  HIP_OK(hipMalloc(&nbd, (size_t)n * sizeof(int32_t)));
  HIP_OK(hipMalloc(&part, blocks * sizeof(double)));
  {
    std::vector<double> h((size_t)n, -1.0);
    HIP_OK(hipMemcpy(l, h.data(), vb, hipMemcpyHostToDevice));
    std::fill(h.begin(), h.end(), 1.0);
    HIP_OK(hipMemcpy(u, h.data(), vb, hipMemcpyHostToDevice));
    std::vector<int32_t> nb((size_t)n, 2);
    HIP_OK(hipMemcpy(nbd, nb.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(x[0], 0, vb));   // x0 = 0 in the FIRST pair
  }

  char task[60], csave[60];
  int32_t lsave[4] = {0, 0, 0, 0}, isave[44], cur = 0;
  double dsave[29], f = 0.0;
  std::memset(isave, 0, sizeof isave);
  std::memset(dsave, 0, sizeof dsave);
  std::memset(task, ' ', 60);
  std::memset(csave, ' ', 60);
  std::memcpy(task, "START", 5);
  std::vector<double> hpart(blocks);
  for (;;) {
    rc = lbfgsb_hip_setulb_dev_pp(ctx, x[0], x[1], l, u, nbd, &f, g[0], g[1], /*factr*/ 0.0, /*pgtol*/ 0.0, task,
                                  /*iprint*/ -1, csave, lsave, isave, dsave, &cur);
    if (rc != LBFGSB_OK) {
      std::fprintf(stderr, "lbfgsb_hip_setulb_dev_pp: %d (%s)\n", rc, lbfgsb_hip_last_error());
      return 1;
    }
    if (!std::strncmp(task, "FG", 2)) {
      // evaluate AT x[cur] INTO g[cur], on the solver's stream (ordered behind the pass that wrote x[cur])
      hipLaunchKernelGGL(objective, dim3(blocks), dim3(256), 0, stream, n, x[cur], g[cur], part);
      HIP_OK(hipMemcpyAsync(hpart.data(), part, blocks * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      double s = 0.0;
      for (double v : hpart) s += v;
      f = 0.5 * s;
    } else if (!std::strncmp(task, "NEW_X", 5)) {
      std::printf("iterate %3d  nfg %3d  nseg %9d  nfree %9d  f = %.12e  |proj g| = %.4e\n", (int)isave[29],
                  (int)isave[33], (int)isave[32], (int)isave[37], f, dsave[12]);
      if (isave[29] >= max_iter) {
        std::memcpy(task, "STOP: ITERATION LIMIT", 21);   // driver2.f90's way of ending a run
        std::memset(task + 21, ' ', 39);
      }
    } else {
      break;
    }
  }
  std::printf("task = %.48s\n", task);
  std::printf("iterations = %d  nfg = %d  f = %.16e\n", (int)isave[29], (int)isave[33], f);
  lbfgsb_hip_destroy(ctx);
  for (int k = 0; k < 2; ++k) (void)hipFree(x[k]), (void)hipFree(g[k]);
  (void)hipFree(l), (void)hipFree(u), (void)hipFree(nbd), (void)hipFree(part);
  return 0;
}
