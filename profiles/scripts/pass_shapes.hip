// Micro-benchmark for the shape of the library's passes over W (n = 1e8 rows, fp64):
//   NR read streams + NW write streams, 16 B per lane, as in write_cost.hip, in these forms:
//   base      grid-stride loop, loads -> compute -> stores per trip (what round 1 shipped)
//   pipe      software-pipelined: the NEXT trip's loads are issued BEFORE this trip's stores, so
//             the wait for those loads is a counted vmcnt(NW) and never waits for a store
//             (CDNA4's vmcnt counts stores too: a loop that stores stalls on its own stores)
//   flat      one trip per workgroup (grid = n / 512), no loop at all
//   tiled     the 20 "W columns" live in ONE array tiled [row block of 512][column][512 rows], so
//             a workgroup trip reads one contiguous 80 KB region instead of 20 separate streams
//             (4 n-vectors stay separate streams, like x, g, l, u)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 pass_shapes.hip -o pass_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d2 ldnt(const double *p) { return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p)); }
__device__ __forceinline__ void stnt(double *p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p)); }

// address of "column" j for row pair iv (rows 2iv, 2iv+1)
template <bool TILED, int NCOL>
__device__ __forceinline__ const double *caddr(const double *w, const double *vec, int64_t ld, int j, int64_t iv) {
  if constexpr (TILED) {
    if (j < NCOL) return w + (iv >> 8) * (int64_t)(NCOL * 512) + j * 512 + ((iv & 255) << 1);
    return vec + (int64_t)(j - NCOL) * ld + iv * 2;
  } else {
    return w + (int64_t)j * ld + iv * 2;
  }
}

template <int NR, int NW, bool TILED>
__global__ __launch_bounds__(256) void k_base(int64_t n, const double *__restrict__ w, const double *__restrict__ vec,
                                              double *out, int64_t ld, double *sink) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  d2 acc = {0.0, 0.0};
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    d2 v[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = ldnt(caddr<TILED, 20>(w, vec, ld, j, iv));
    d2 s = {1.0, 2.0};
#pragma unroll
    for (int j = 0; j < NR; ++j) s += v[j];
    acc += s;
#pragma unroll
    for (int j = 0; j < NW; ++j) stnt(out + (int64_t)j * ld + iv * 2, s + (double)j);
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

template <int NR, int NW, bool TILED>
__global__ __launch_bounds__(256) void k_pipe(int64_t n, const double *__restrict__ w, const double *__restrict__ vec,
                                              double *out, int64_t ld, double *sink) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  d2 acc = {0.0, 0.0};
  int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x;
  d2 v[NR];
  if (iv < nv) {
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = ldnt(caddr<TILED, 20>(w, vec, ld, j, iv));
  }
  while (iv < nv) {
    d2 s = {1.0, 2.0};
#pragma unroll
    for (int j = 0; j < NR; ++j) s += v[j];
    acc += s;
    const int64_t nx = iv + stride;
    const int64_t nxc = nx < nv ? nx : iv;   // past the end: re-read this trip's rows (no branch)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = ldnt(caddr<TILED, 20>(w, vec, ld, j, nxc));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NW; ++j) stnt(out + (int64_t)j * ld + iv * 2, s + (double)j);
    __builtin_amdgcn_sched_barrier(0);
    iv = nx;
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}


// pipe with hand-placed waits: loads are inline asm (the compiler does not track them), the wait
// for them is a counted s_waitcnt vmcnt(NW) AFTER this trip's stores were issued, so the stores
// of a trip are never waited for inside the loop (the compiler's own wait placement merges the
// loop-entry state with the back-edge state and ends in vmcnt(0) at the loop header)
__device__ __forceinline__ d2 ldnt_asm(const double *p) {
  d2 v;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin(d2 &v) { asm volatile("" : "+v"(v)); }

template <int NR, int NW, bool TILED>
__global__ __launch_bounds__(256) void k_pipe_asm(int64_t n, const double *__restrict__ w, const double *__restrict__ vec,
                                                  double *out, int64_t ld, double *sink) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  d2 acc = {0.0, 0.0};
  int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (iv >= nv) return;
  d2 v[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) v[j] = ldnt_asm(caddr<TILED, 20>(w, vec, ld, j, iv));
  wait_vm<0>();
#pragma unroll
  for (int j = 0; j < NR; ++j) pin(v[j]);
  while (iv < nv) {
    d2 s = {1.0, 2.0};
#pragma unroll
    for (int j = 0; j < NR; ++j) s += v[j];
    acc += s;
    const int64_t nx = iv + stride;
    const int64_t nxc = nx < nv ? nx : iv;
    pin(s);   // the sums are complete before the operand registers are overwritten
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = ldnt_asm(caddr<TILED, 20>(w, vec, ld, j, nxc));
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      d2 o = s + (double)j;
      asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(out + (int64_t)j * ld + iv * 2), "v"(o) : "memory");
    }
    wait_vm<NW>();   // every load has landed; the NW stores may still be in flight
#pragma unroll
    for (int j = 0; j < NR; ++j) pin(v[j]);
    iv = nx;
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

template <int NR, int NW, bool TILED>
__global__ __launch_bounds__(256) void k_flat(int64_t n, const double *__restrict__ w, const double *__restrict__ vec,
                                              double *out, int64_t ld, double *sink) {
  const int64_t nv = n / 2;
  const int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (iv >= nv) return;
  d2 v[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) v[j] = ldnt(caddr<TILED, 20>(w, vec, ld, j, iv));
  d2 s = {1.0, 2.0};
#pragma unroll
  for (int j = 0; j < NR; ++j) s += v[j];
#pragma unroll
  for (int j = 0; j < NW; ++j) stnt(out + (int64_t)j * ld + iv * 2, s + (double)j);
  if (s.x + s.y == 12345.678) sink[0] = s.x;
}


// ---- round 5: the ONE-store-stream candidate of the storing pass, as exactly as a bare
// kernel can state it.  Today (k_cols): 2 col W columns + x + g read, the subspace step formed per row, THREE
// streams written (trial x, the pending pair's Ws / Wy column), four sums reduced.  Candidate (k_ring): the
// library keeps rings of col + 1 ITERATES and col + 1 GRADIENTS instead of Ws / Wy; y_j = G_{j+1} - G_j is
// always the stored y bit for bit, s_j = X_{j+1} - X_j is the stored s only after a unit step
// (src/lbfgsb.f90:822, 2313-2314), so each s column reads two ADDRESS-SELECTED operands: (X_{j+1}, X_j), or
// (materialised Ws_j, a zero buffer) for a column whose step was not 1.  One stream written (the trial x =
// the next iterate of the ring).  Same arithmetic per row, same reductions.
struct RingArgs {
  const double *a[10], *b[10];   // s_j = a_j - b_j
  int bstride[10];               // 2 (a ring operand: advances with the rows) or 0 (the zero buffer)
  const double *g[11];           // gradient ring
  double cs[10], cy[10];
};
__device__ __forceinline__ void block_sums4(double (&acc)[4], double *part) {
  __shared__ double sm[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}
__global__ __launch_bounds__(256) void k_ring(int64_t n, RingArgs A, const signed char *__restrict__ iwhere, double *xout,
                                              double *part) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  double acc[4] = {0, 0, 0, 0};
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    d2 av[10], bv[10], gv[11];
#pragma unroll
    for (int j = 0; j < 10; ++j) av[j] = ldnt(A.a[j] + iv * 2);
#pragma unroll
    for (int j = 0; j < 10; ++j) bv[j] = ldnt(A.b[j] + iv * A.bstride[j]);
#pragma unroll
    for (int j = 0; j < 11; ++j) gv[j] = ldnt(A.g[j] + iv * 2);
    const short iw2 = *reinterpret_cast<const short *>(iwhere + iv * 2);
    d2 dk = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const d2 sj = av[j] - bv[j], yj = gv[j + 1] - gv[j];
      dk += sj * A.cs[j] + yj * A.cy[j];
    }
    if ((signed char)(iw2 & 0xff) > 0) dk.x = 0.0;
    if ((signed char)(iw2 >> 8) > 0) dk.y = 0.0;
    const d2 xv = av[9];   // the newest iterate is the current x
    const d2 zv = xv + dk;
    acc[0] += dk.x * gv[10].x + dk.y * gv[10].y;
    acc[1] += dk.x * dk.x + dk.y * dk.y;
    acc[2] += (zv.x == 1.0) + (zv.y == 1.0);
    acc[3] += zv.x + zv.y;
    stnt(xout + iv * 2, zv);
  }
  block_sums4(acc, part);
}
struct ColArgs {
  const double *wy[10], *ws[10];
  const double *x, *g;
  double cs[10], cy[10];
};
template <int NW>
__global__ __launch_bounds__(256) void k_cols(int64_t n, ColArgs A, const signed char *__restrict__ iwhere, double *xout,
                                              double *cwy, double *cws, double *part) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  double acc[4] = {0, 0, 0, 0};
  for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += stride) {
    d2 yv[10], sv[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) yv[j] = ldnt(A.wy[j] + iv * 2);
#pragma unroll
    for (int j = 0; j < 10; ++j) sv[j] = ldnt(A.ws[j] + iv * 2);
    const d2 xv = ldnt(A.x + iv * 2), gv = ldnt(A.g + iv * 2);
    const short iw2 = *reinterpret_cast<const short *>(iwhere + iv * 2);
    d2 dk = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 10; ++j) dk += sv[j] * A.cs[j] + yv[j] * A.cy[j];
    if ((signed char)(iw2 & 0xff) > 0) dk.x = 0.0;
    if ((signed char)(iw2 >> 8) > 0) dk.y = 0.0;
    const d2 zv = xv + dk;
    acc[0] += dk.x * gv.x + dk.y * gv.y;
    acc[1] += dk.x * dk.x + dk.y * dk.y;
    acc[2] += (zv.x == 1.0) + (zv.y == 1.0);
    acc[3] += zv.x + zv.y;
    stnt(xout + iv * 2, zv);
    if (NW >= 3) {
      stnt(cwy + iv * 2, yv[9] + gv);
      stnt(cws + iv * 2, sv[9] + xv);
    }
  }
  block_sums4(acc, part);
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

template <int NR, int NW, bool TILED>
void suite(int64_t n, const double *w, const double *vec, double *out, double *sink) {
  const double gb = (NR + NW) * 8.0 * n / 1e9;
  const char *lay = TILED ? "tiled " : "column";
  for (int grid : {512, 768, 1024, 2048}) {
    float ms = timeit([&] { hipLaunchKernelGGL((k_base<NR, NW, TILED>), dim3(grid), dim3(256), 0, 0, n, w, vec, out, n, sink); });
    printf("reads %2d writes %d %s base grid %6d  %7.3f ms %7.1f GB/s\n", NR, NW, lay, grid, ms, gb / ms * 1e3);
    if (NW > 0) {
      ms = timeit([&] { hipLaunchKernelGGL((k_pipe<NR, NW, TILED>), dim3(grid), dim3(256), 0, 0, n, w, vec, out, n, sink); });
      printf("reads %2d writes %d %s pipe grid %6d  %7.3f ms %7.1f GB/s\n", NR, NW, lay, grid, ms, gb / ms * 1e3);
      ms = timeit([&] { hipLaunchKernelGGL((k_pipe_asm<NR, NW, TILED>), dim3(grid), dim3(256), 0, 0, n, w, vec, out, n, sink); });
      printf("reads %2d writes %d %s pasm grid %6d  %7.3f ms %7.1f GB/s\n", NR, NW, lay, grid, ms, gb / ms * 1e3);
    }
  }
  const int gflat = (int)((n / 2 + 255) / 256);
  const float ms = timeit([&] { hipLaunchKernelGGL((k_flat<NR, NW, TILED>), dim3(gflat), dim3(256), 0, 0, n, w, vec, out, n, sink); });
  printf("reads %2d writes %d %s flat grid %6d  %7.3f ms %7.1f GB/s\n", NR, NW, lay, gflat, ms, gb / ms * 1e3);
  fflush(stdout);
}


// round 5 (mode i): would an INTERLEAVED pair layout -- (y, s) of a row side by side, 16 bytes per row and column:
// col fat read streams instead of 2 col thin ones, ONE fat store stream for the committed pair instead of two --
// change the storing pass?  Same bytes either way.  NT thin read streams (8 B/row: lane = 16 B for 2 rows), NF fat ones
// (16 B/row: lane = 32 B), WT thin / WF fat write streams; pipelined like k_pipe.
template <int NT_, int NF, int WT, int WF>
__global__ __launch_bounds__(256) void k_inter(int64_t n, const double *__restrict__ w, double *out, double *sink) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  d2 acc = {0.0, 0.0};
  int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x;
  d2 v[NT_ + 2 * NF];
  auto issue = [&](int64_t i) {
#pragma unroll
    for (int j = 0; j < NT_; ++j) v[j] = ldnt(w + (int64_t)j * n + i * 2);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const double *b = w + (int64_t)NT_ * n + (int64_t)j * 2 * n + i * 4;
      v[NT_ + 2 * j] = ldnt(b), v[NT_ + 2 * j + 1] = ldnt(b + 2);
    }
  };
  if (iv < nv) issue(iv);
  while (iv < nv) {
    d2 s = {1.0, 2.0};
#pragma unroll
    for (int j = 0; j < NT_ + 2 * NF; ++j) s += v[j];
    acc += s;
    const int64_t nx = iv + stride;
    const int64_t nxc = nx < nv ? nx : iv;
    __builtin_amdgcn_sched_barrier(0);
    issue(nxc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < WT; ++j) stnt(out + (int64_t)j * n + iv * 2, s + (double)j);
#pragma unroll
    for (int j = 0; j < WF; ++j) {
      double *b = out + (int64_t)WT * n + (int64_t)j * 2 * n + iv * 4;
      stnt(b, s), stnt(b + 2, s + 1.0);
    }
    __builtin_amdgcn_sched_barrier(0);
    iv = nx;
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}


// round 5 (mode j): the interleaved pair layout with ONE row per lane -- a thin stream is an 8-byte load per lane
// (what the headline's read-only pass does today: V = 1), a fat stream ((y, s) of a row side by side) a DENSE 16-byte
// load per lane.  PIPE: two trips in flight as in for_rows_raw.
__device__ __forceinline__ double ldnt1(const double *p) { return __builtin_nontemporal_load(p); }
template <int NT_, int NF, int WT, int WF>
__global__ __launch_bounds__(256) void k_inter1(int64_t n, const double *__restrict__ w, double *out, double *sink) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  double acc = 0.0;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double v[NT_];
  d2 f[NF > 0 ? NF : 1];
  auto issue = [&](int64_t r) {
#pragma unroll
    for (int j = 0; j < NT_; ++j) v[j] = ldnt1(w + (int64_t)j * n + r);
#pragma unroll
    for (int j = 0; j < NF; ++j) f[j] = ldnt(w + (int64_t)NT_ * n + (int64_t)j * 2 * n + r * 2);
  };
  if (i < n) issue(i);
  while (i < n) {
    double s = 1.0;
#pragma unroll
    for (int j = 0; j < NT_; ++j) s += v[j];
#pragma unroll
    for (int j = 0; j < NF; ++j) s += f[j].x * 0.5 + f[j].y;
    acc += s;
    const int64_t nx = i + stride;
    const int64_t nxc = nx < n ? nx : i;
    __builtin_amdgcn_sched_barrier(0);
    issue(nxc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < WT; ++j) __builtin_nontemporal_store(s + (double)j, out + (int64_t)j * n + i);
#pragma unroll
    for (int j = 0; j < WF; ++j) {
      d2 o = {s, s + 1.0};
      stnt(out + (int64_t)WT * n + (int64_t)j * 2 * n + i * 2, o);
    }
    __builtin_amdgcn_sched_barrier(0);
    i = nx;
  }
  if (acc == 12345.678) sink[0] = acc;
}


// round 5 (mode p): would separating reads and writes in TIME help?  Persistent grid (every workgroup resident),
// T trips of loads with the results kept in registers, a grid barrier, T trips of stores, a grid barrier.  BAR = false:
// the same loop without the barriers (each workgroup alternates on its own) -- the cost of the restructuring alone.
__device__ __forceinline__ void grid_barrier(unsigned int *ctr, unsigned int target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // (bounded: a grid that is not fully resident must not hang the card -- the timing is then meaningless, and
    //  the host checks residency before it launches)
    for (int spins = 0; spins < 4000000 && __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins)
      __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
}
template <int NR, int NW, int T, bool BAR>
__global__ __launch_bounds__(256) void k_phase(int64_t n, const double *__restrict__ w, double *out, double *sink,
                                               unsigned int *ctr) {
  const int64_t nv = n / 2, stride = (int64_t)gridDim.x * 256;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  d2 acc = {0.0, 0.0};
  unsigned int phase = 0;
  for (int64_t base = 0; base < nv; base += stride * T) {   // (uniform trip count: every workgroup reaches every barrier)
    d2 r[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t iv = base + t * stride + t0;
      d2 sum = {1.0, 2.0};
      if (iv < nv) {
        d2 v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = ldnt(w + (int64_t)j * n + iv * 2);
#pragma unroll
        for (int j = 0; j < NR; ++j) sum += v[j];
      }
      r[t] = sum;
      acc += sum;
    }
    if (BAR) grid_barrier(ctr, ++phase * gridDim.x);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t iv = base + t * stride + t0;
      if (iv < nv) {
#pragma unroll
        for (int j = 0; j < NW; ++j) stnt(out + (int64_t)j * n + iv * 2, r[t] + (double)j);
      }
    }
    if (BAR) grid_barrier(ctr, ++phase * gridDim.x);
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

int main(int argc, char **argv) {
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int64_t n = 100000000;   // multiple of 512
  double *w, *out, *sink;
  CK(hipMalloc(&w, (size_t)n * 24 * 8));
  CK(hipMalloc(&out, (size_t)n * 14 * 8));   // (mode c: up to 10 materialised columns behind 3 store targets)
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(w, 0, (size_t)n * 24 * 8));
  CK(hipMemset(out, 0, (size_t)n * 14 * 8));
  const double *vec = w + (size_t)n * 20;   // the 4 separate n-vectors of the tiled form
  if (argc > 1 && argv[1][0] == 'c') {
    // round 5: today's storing pass against the one-store-stream candidate, both with their reductions
    signed char *iw;
    double *part, *zero;
    CK(hipMalloc(&iw, (size_t)n));
    CK(hipMemset(iw, 0, (size_t)n));
    CK(hipMalloc(&part, 4 * 4096 * 8));
    CK(hipMalloc(&zero, 256));
    CK(hipMemset(zero, 0, 256));
    ColArgs C{};
    for (int j = 0; j < 10; ++j) C.wy[j] = w + (size_t)n * j, C.ws[j] = w + (size_t)n * (10 + j), C.cs[j] = 0.5 + j, C.cy[j] = 0.25 * j;
    C.x = w + (size_t)n * 20, C.g = w + (size_t)n * 21;
    for (int pass = 0; pass < 2; ++pass) {
      for (int grid : {512, 768, 1024}) {
        float ms = timeit([&] { hipLaunchKernelGGL((k_cols<3>), dim3(grid), dim3(256), 0, 0, n, C, iw, out, out + n, out + 2 * n, part); });
        printf("today     22 read streams, 3 written, 4 sums            grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_cols<1>), dim3(grid), dim3(256), 0, 0, n, C, iw, out, out + n, out + 2 * n, part); });
        printf("(bound)   22 read streams, 1 written, 4 sums            grid %5d  %7.3f ms\n", grid, ms);
        for (int nmat : {0, 1, 3, 10}) {
          // iterates X_0 .. X_10 = streams 0..10, gradients = streams 11..21, materialised Ws_j = out + (3 + k) n
          RingArgs R{};
          int k = 0;
          for (int j = 0; j < 10; ++j) {
            const bool mat = j < nmat;
            R.a[j] = mat ? out + (size_t)n * (3 + k++) : w + (size_t)n * (j + 1);
            R.b[j] = mat ? zero : w + (size_t)n * j;
            R.bstride[j] = mat ? 0 : 2;
            R.cs[j] = 0.5 + j, R.cy[j] = 0.25 * j;
          }
          if (nmat == 10) R.a[9] = w + (size_t)n * 10;   // (the current x is always a ring operand)
          for (int j = 0; j < 11; ++j) R.g[j] = w + (size_t)n * (11 + j);
          ms = timeit([&] { hipLaunchKernelGGL(k_ring, dim3(grid), dim3(256), 0, 0, n, R, iw, out, part); });
          printf("candidate %2d + 11 ring streams, %2d materialised s columns, 1 written, 4 sums  grid %5d  %7.3f ms\n",
                 11, nmat, grid, ms);
        }
      }
      printf("\n");
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'p') {
    unsigned int *ctr;
    CK(hipMalloc(&ctr, 64));
    for (int pass = 0; pass < 2; ++pass) {
      for (int grid : {512, 768}) {
        float ms = timeit([&] { hipLaunchKernelGGL((k_inter<22, 0, 3, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("storing pass today (pipelined loop)                        grid %5d  %7.3f ms\n", grid, ms);
#define PH(T, BAR) \
        { int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)&k_phase<22, 3, T, BAR>, 256, 0)); \
          if (per_cu * 256 < grid) { printf("phases of %2d trips: only %d workgroups per CU resident, grid %d skipped\n", T, per_cu, grid); } else { \
        ms = timeit([&] { CK(hipMemsetAsync(ctr, 0, 64, 0)); hipLaunchKernelGGL((k_phase<22, 3, T, BAR>), dim3(grid), dim3(256), 0, 0, n, w, out, sink, ctr); }); \
        printf("phases of %2d trips, %s                  grid %5d  %7.3f ms\n", T, BAR ? "grid barriers between reads and writes" : "no barrier (restructuring only)      ", grid, ms); } }
        PH(4, false) PH(4, true) PH(8, false) PH(8, true) PH(16, false) PH(16, true)
#undef PH
      }
      printf("\n");
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'j') {
    for (int pass = 0; pass < 2; ++pass) {
      for (int grid : {512, 768, 1024, 2048}) {
        float ms = timeit([&] { hipLaunchKernelGGL((k_inter1<22, 0, 0, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("read-only pass, one row per lane, today:       22 thin (8 B) loads                     grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_inter1<4, 9, 0, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("read-only pass, one row per lane, interleaved:  4 thin + 9 fat (16 B) loads            grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_inter1<22, 0, 3, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("storing pass,   one row per lane, columns:     22 thin loads, 3 thin stores            grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_inter1<4, 9, 1, 1>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("storing pass,   one row per lane, interleaved:  4 thin + 9 fat loads, 1 thin + 1 fat store grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_inter<22, 0, 3, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("storing pass,   two rows per lane, today:      22 x 16 B loads, 3 x 16 B stores        grid %5d  %7.3f ms\n", grid, ms);
      }
      printf("\n");
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'i') {
    for (int pass = 0; pass < 2; ++pass) {
      for (int grid : {512, 768, 1024}) {
        float ms = timeit([&] { hipLaunchKernelGGL((k_inter<22, 0, 3, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("storing pass, today:       22 thin reads,           3 thin writes            grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_inter<2, 10, 1, 1>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("storing pass, interleaved:  2 thin + 10 fat reads,  1 thin + 1 fat write     grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_inter<22, 0, 0, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("read-only pass, today:     22 thin reads                                     grid %5d  %7.3f ms\n", grid, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_inter<2, 10, 0, 0>), dim3(grid), dim3(256), 0, 0, n, w, out, sink); });
        printf("read-only pass, interleaved: 2 thin + 10 fat reads                           grid %5d  %7.3f ms\n", grid, ms);
      }
      printf("\n");
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 's') {
    // round 4: what would the storing pass gain with ONE store stream instead of three
    // (s_j, y_j formed in registers from a ring of iterates / gradients instead of stored)?  The bare mixes:
    // 22 reads + 3 / 2 / 1 writes -- and 23 reads + 1 write, the mix a ring would really have (col + 1
    // iterates and gradients instead of col columns each: one read stream more)
    for (int pass = 0; pass < 2; ++pass) {
      suite<22, 3, false>(n, w, vec, out, sink);
      suite<22, 2, false>(n, w, vec, out, sink);
      suite<22, 1, false>(n, w, vec, out, sink);
      suite<24, 1, false>(n, w, vec, out, sink);
      printf("\n");
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'r') {
    // round 3: the stream mix of the iteration's two passes as they are now (uniform bounds, ping-pong entry):
    // storing pass = 22 fp64 read streams (+ 1 B/row of iwhere) and 3 write streams, read-only pass = 22 read streams
    for (int pass = 0; pass < 2; ++pass) {
      suite<22, 3, false>(n, w, vec, out, sink);
      suite<22, 5, false>(n, w, vec, out, sink);
      suite<22, 0, false>(n, w, vec, out, sink);
      printf("\n");
    }
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    suite<24, 0, false>(n, w, vec, out, sink);
    suite<24, 0, true>(n, w, vec, out, sink);
    suite<24, 7, false>(n, w, vec, out, sink);
    suite<24, 7, true>(n, w, vec, out, sink);
    suite<22, 0, false>(n, w, vec, out, sink);
    suite<22, 0, true>(n, w, vec, out, sink);
    printf("\n");
  }
  return 0;
}
