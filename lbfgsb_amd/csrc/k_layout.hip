// k_layout.hip -- the tile-local free-row layout of W ("compact W")
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
//
// The reference's cmprlb, subsm and formk loops run over the free variables through Index (src/lbfgsb.f90:1565-1583,
// :2743-2778, :1756-1793; freev builds the list, :2044-2054).  Streaming all n rows of the 2m columns under a mask
// moves n/nfree times the bytes those loops define.  Under this layout every column keeps, inside each aligned tile
// of CW_TILE = 128 rows, the rows whose layout bit is set FIRST (ascending: Index order), the others behind them, so
// the two passes over W read one contiguous run of about 128 nfree/n elements per column and tile.  The bits are the
// free set at the time the layout was made; rows that changed status since are still found (at their slot behind
// the run), so a stale layout costs bytes, never correctness.  Row ownership, x, g, l, u, iwhere stay untouched.
#include "kernels_common.hpp"

namespace lbk {

__global__ __launch_bounds__(BLOCK) void lmask_ones_kernel(int64_t n, uint64_t *__restrict__ lmask, int64_t nwords) {
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r0 = w * 64;
    uint64_t m = 0;
    if (r0 + 64 <= n) m = ~0ull;
    else if (r0 < n) m = (1ull << (n - r0)) - 1ull;
    lmask[w] = m;
  }
}
void launch_lmask_ones(Queue &q, int64_t n, uint64_t *lmask) {
  const int64_t nwords = ((n + CW_TILE - 1) / CW_TILE) * (CW_TILE / 64);
  int gr = (int)std::min<int64_t>((nwords + BLOCK - 1) / BLOCK, MAX_BLOCKS);
  if (gr < 1) gr = 1;
  hipLaunchKernelGGL(lmask_ones_kernel, dim3(gr), dim3(BLOCK), 0, q.stream, n, lmask, nwords);
  LB_LAUNCHED(q);
}

// One wave per tile and trip; lane l owns rows l and l + 64 of the tile, as in for_tiles_cw.  A tile whose bits
// change has every live column permuted in place: all of a column's 128 entries are read (old slots) before any
// is written (new slots) -- the wait between the two is explicit, the stores of one lane overwrite what another
// lane has just read.  CB columns per batch keep 4 CB loads in flight.
template <typename T, int CB>
__global__ __launch_bounds__(BLOCK) void w_relayout_kernel(int64_t n, const iw_t *__restrict__ iwhere,
                                                           uint64_t *__restrict__ lmask, T *ws, T *wy, int64_t ldw,
                                                           int m, int head, int col) {
  const int lane = threadIdx.x & 63;
  const int64_t ntile = (n + CW_TILE - 1) / CW_TILE;
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tr = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tr < ntile; tr += stride) {
    const int64_t tb = tr << 7, i0 = tb + lane, i1 = i0 + 64;
    const uint64_t o0 = lmask[2 * tr], o1 = lmask[2 * tr + 1];
    bool w0 = i0 < n, w1 = i1 < n;
    if (iwhere) {
      w0 = w0 && iwhere[w0 ? i0 : 0] <= 0;
      w1 = w1 && iwhere[w1 ? i1 : 0] <= 0;
    }
    const uint64_t n0 = __ballot(w0), n1 = __ballot(w1);
    if (n0 == o0 && n1 == o1) continue;  // (wave-uniform)
    auto slots = [&](uint64_t m0, uint64_t m1, int64_t &s0, int64_t &s1) {
      const int c0 = __popcll(m0), tf = c0 + __popcll(m1);
      const uint64_t below = (1ull << lane) - 1ull;
      const int b0 = __popcll(m0 & below), b1 = c0 + __popcll(m1 & below);
      s0 = tb + (((m0 >> lane) & 1ull) ? b0 : tf + (lane - b0));
      s1 = tb + (((m1 >> lane) & 1ull) ? b1 : tf + (64 + lane - b1));
    };
    int64_t so0, so1, sn0, sn1;
    slots(o0, o1, so0, so1);
    slots(n0, n1, sn0, sn1);
    for (int j0 = 0; j0 < col; j0 += CB) {
      T vy[CB][2], vs[CB][2];
#pragma unroll
      for (int jj = 0; jj < CB; ++jj) {
        const int j = j0 + jj < col ? j0 + jj : col - 1;
        const int64_t off = (int64_t)((head - 1 + j) % m) * ldw;
        vy[jj][0] = wy[off + so0], vy[jj][1] = wy[off + so1];
        vs[jj][0] = ws[off + so0], vs[jj][1] = ws[off + so1];
      }
      // every load of the batch has landed in every lane before the first store goes out
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int jj = 0; jj < CB; ++jj) {
        if (j0 + jj < col) {
          const int64_t off = (int64_t)((head - 1 + (j0 + jj)) % m) * ldw;
          wy[off + sn0] = vy[jj][0], wy[off + sn1] = vy[jj][1];
          ws[off + sn0] = vs[jj][0], ws[off + sn1] = vs[jj][1];
        }
      }
      // ... and the stores are out before the next batch reads the same tile of OTHER columns (no overlap: a
      // batch touches its own columns only), so no wait is needed here
    }
    if (lane == 0) lmask[2 * tr] = n0, lmask[2 * tr + 1] = n1;
  }
}
template <typename T>
void launch_w_relayout(Queue &q, int64_t n, const iw_t *iwhere, uint64_t *lmask, WStore<T> w, int head, int col) {
  const int64_t ntile = (n + CW_TILE - 1) / CW_TILE;
  int gr = (int)std::min<int64_t>((ntile + 3) / 4, 2048);
  if (gr < 1) gr = 1;
  hipLaunchKernelGGL((w_relayout_kernel<T, 4>), dim3(gr), dim3(BLOCK), 0, q.stream, n, iwhere, lmask, w.ws, w.wy,
                     w.ld, w.m, head, col);
  LB_LAUNCHED(q);
}

template void launch_w_relayout<double>(Queue &, int64_t, const iw_t *, uint64_t *, WStore<double>, int, int);
template void launch_w_relayout<float>(Queue &, int64_t, const iw_t *, uint64_t *, WStore<float>, int, int);

}  // namespace lbk
