// k_freev.hip -- freev: counts of the free set, the changed-row list, the mirror of Index / Indx2
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

namespace lbk {

// =========================== freev (:1980-2059) ==============================
// sums: [0] nfree, [1] nenter, [2] nleave, [3] rows whose status changed (= the length of the list, as a
// sum: the host needs no copy of the list's atomic position counter).  chg_count / cnt_zero: two position
// counters used in turn -- this launch appends through chg_count and zeroes the OTHER one for the next
// launch (no memset command in front of the kernel).
// WRITE = false: counts and the list only, wasfree untouched -- the pass run SPECULATIVELY behind the
// evaluation of a trial point (solver.hip, phase_entry); if its result is used, freev_apply_kernel brings
// wasfree up to date from the list
template <bool WRITE>
__global__ __launch_bounds__(BLOCK) void freev_count_kernel(int64_t n,
                                                            const iw_t *__restrict__ iwhere,
                                                            int8_t *wasfree, double *part,
                                                            uint32_t *chg, uint32_t chg_cap,
                                                            uint32_t *chg_count, uint32_t *cnt_zero) {
  // 16 rows per lane and trip (one 16-byte load of each byte array).  Rows whose status changed
  // are collected per workgroup in LDS and appended to the global list with ONE global atomic
  // per flush (a same-address atomic per row would serialise: 1e5 changes x ~12 ns)
  constexpr int R = 16, LCAP = 8192;
  __shared__ uint32_t lbuf[LCAP];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cnt_zero = 0;
  __syncthreads();
  double acc[4] = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * R;
  const int64_t ntrip = (n + stride - 1) / stride;  // uniform trip count (barriers inside)
  for (int64_t trip = 0; trip < ntrip; ++trip) {
    const int64_t i = trip * stride + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * R;
    if (i < n) {
      typedef int v4i __attribute__((ext_vector_type(4)));
      union {
        v4i v;
        int8_t b[R];
      } iw, wf;
      const bool full = i + R <= n;  // (both arrays are allocated with 32 spare elements, but
                                     //  rows beyond n must neither be counted nor written)
      if (full) {
        iw.v = *reinterpret_cast<const v4i *>(iwhere + i);
        wf.v = *reinterpret_cast<const v4i *>(wasfree + i);
      } else {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          iw.b[k] = i + k < n ? iwhere[i + k] : (iw_t)1;
          wf.b[k] = i + k < n ? wasfree[i + k] : (int8_t)0;
        }
      }
      int nfr = 0, nen = 0, nlv = 0;
      unsigned changed = 0;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const bool fr = iw.b[k] <= 0, was = wf.b[k] != 0;
        nfr += fr, nen += fr && !was, nlv += !fr && was;
        changed |= (fr != was) ? (1u << k) : 0u;
        wf.b[k] = fr ? 1 : 0;
      }
      acc[0] += nfr, acc[1] += nen, acc[2] += nlv, acc[3] += __builtin_popcount(changed);
      if (changed) {  // (few rows: keeps the pass that follows free of drained store traffic)
        if (chg) {
          const uint32_t pos = atomicAdd(&lcount, (uint32_t)__builtin_popcount(changed));  // LDS atomic
          uint32_t w = pos;
#pragma unroll
          for (int k = 0; k < R; ++k)
            if ((changed >> k) & 1u) lbuf[w++] = (uint32_t)(i + k) | (wf.b[k] ? 0u : 0x80000000u);
        }
        if (!WRITE) {
          // (speculative pass: nothing stored)
        } else if (full) {
          *reinterpret_cast<v4i *>(wasfree + i) = wf.v;
        } else {
#pragma unroll
          for (int k = 0; k < R; ++k)
            if (i + k < n) wasfree[i + k] = wf.b[k];
        }
      }
    }
    if (chg) {
      __syncthreads();
      const uint32_t cnt = lcount;
      if (cnt > LCAP - BLOCK * R || trip == ntrip - 1) {  // uniform: flush
        if (threadIdx.x == 0) gbase = cnt ? atomicAdd(chg_count, cnt) : 0u;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt; k += BLOCK)
          if (gbase + k < chg_cap) chg[gbase + k] = lbuf[k];
        __syncthreads();
        if (threadIdx.x == 0) lcount = 0;
        __syncthreads();
      }
    }
  }
  block_reduce_store<4>(acc, 4, 0, 0, part, MAX_BLOCKS);
}
void launch_freev_count(Queue &q, int64_t n, const iw_t *iwhere, int8_t *wasfree, uint32_t *chg,
                        uint32_t chg_cap, uint32_t *cnt2, int parity, int write) {
  const int gr = grid_for(n, 16);
  if (write)
    hipLaunchKernelGGL(freev_count_kernel<true>, dim3(gr), dim3(BLOCK), 0, q.stream, n, iwhere, wasfree,
                       q.part(), chg, chg_cap, cnt2 + (parity & 1), cnt2 + ((parity & 1) ^ 1));
  else
    hipLaunchKernelGGL(freev_count_kernel<false>, dim3(gr), dim3(BLOCK), 0, q.stream, n, iwhere, wasfree,
                       q.part(), chg, chg_cap, cnt2 + (parity & 1), cnt2 + ((parity & 1) ^ 1));
  LB_LAUNCHED(q);
  launch_finalize(q, gr, 4, 0, 0);
}
// wasfree brought up to date from the list of a speculative counting pass (<= cap entries, length on the device)
__global__ __launch_bounds__(BLOCK) void freev_apply_kernel(const uint32_t *__restrict__ chg,
                                                            const uint32_t *__restrict__ cnt_ptr, uint32_t cap,
                                                            int8_t *wasfree) {
  uint32_t cnt = *cnt_ptr;
  if (cnt > cap) cnt = cap;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += gridDim.x * blockDim.x) {
    const uint32_t e = chg[k];
    wasfree[e & 0x7FFFFFFFu] = (e & 0x80000000u) ? 0 : 1;
  }
}
void launch_freev_apply(Queue &q, const uint32_t *chg, const uint32_t *cnt_ptr, uint32_t cap, int8_t *wasfree) {
  hipLaunchKernelGGL(freev_apply_kernel, dim3(8), dim3(BLOCK), 0, q.stream, chg, cnt_ptr, cap, wasfree);
  LB_LAUNCHED(q);
}

// ordered stream compaction reproducing the reference's list orders exactly:
//   Index : free variables ascending from the front, active ascending from the back
//   Indx2 : entering in DESCENDING variable order from the front (the reference walks the
//           old active list, which is stored back to front), leaving ascending from the back.
constexpr int LIST_ITEMS = 4;
constexpr int LIST_CHUNK = BLOCK * LIST_ITEMS;

__device__ __forceinline__ void list_flags(int64_t i, int64_t n, const iw_t *iwhere,
                                           const int8_t *prev, int do_el, int &fr, int &en,
                                           int &lv) {
  fr = en = lv = 0;
  if (i < n) {
    fr = iwhere[i] <= 0;
    if (do_el) {
      const int was = prev[i] != 0;
      en = fr && !was;
      lv = !fr && was;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void list_count_kernel(int64_t n, const iw_t *iwhere,
                                                           const int8_t *prev, int do_el,
                                                           int32_t *tmp) {
  __shared__ int s[3];
  if (threadIdx.x < 3) s[threadIdx.x] = 0;
  __syncthreads();
  int c0 = 0, c1 = 0, c2 = 0;
  for (int k = 0; k < LIST_ITEMS; ++k) {
    int fr, en, lv;
    list_flags((int64_t)blockIdx.x * LIST_CHUNK + threadIdx.x * LIST_ITEMS + k, n, iwhere, prev,
               do_el, fr, en, lv);
    c0 += fr;
    c1 += en;
    c2 += lv;
  }
  atomicAdd(&s[0], c0);
  atomicAdd(&s[1], c1);
  atomicAdd(&s[2], c2);
  __syncthreads();
  if (threadIdx.x < 3) tmp[3 * blockIdx.x + threadIdx.x] = s[threadIdx.x];
}
// exclusive scan of the per-chunk counts (single workgroup); totals in tmp[3*nch ..]
__global__ __launch_bounds__(BLOCK) void list_scan_kernel(int nch, int32_t *tmp) {
  __shared__ int tot[3][BLOCK];
  const int per = (nch + BLOCK - 1) / BLOCK;
  const int b0 = threadIdx.x * per, b1 = min(nch, b0 + per);
  int c[3] = {0, 0, 0};
  for (int b = b0; b < b1; ++b)
    for (int k = 0; k < 3; ++k) c[k] += tmp[3 * b + k];
  for (int k = 0; k < 3; ++k) tot[k][threadIdx.x] = c[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 3; ++k) {
      int run = 0;
      for (int t = 0; t < BLOCK; ++t) {
        const int v = tot[k][t];
        tot[k][t] = run;
        run += v;
      }
      tmp[3 * nch + k] = run;
    }
  }
  __syncthreads();
  int run[3] = {tot[0][threadIdx.x], tot[1][threadIdx.x], tot[2][threadIdx.x]};
  for (int b = b0; b < b1; ++b)
    for (int k = 0; k < 3; ++k) {
      const int v = tmp[3 * b + k];
      tmp[3 * b + k] = run[k];
      run[k] += v;
    }
}
__global__ __launch_bounds__(BLOCK) void list_write_kernel(int64_t n, const iw_t *iwhere,
                                                           const int8_t *prev, int do_el,
                                                           const int32_t *tmp, int nch,
                                                           int32_t *index, int32_t *indx2) {
  __shared__ int sc[3][BLOCK];
  int fr[LIST_ITEMS], en[LIST_ITEMS], lv[LIST_ITEMS];
  int c[3] = {0, 0, 0};
  const int64_t i0 = (int64_t)blockIdx.x * LIST_CHUNK + threadIdx.x * LIST_ITEMS;
  for (int k = 0; k < LIST_ITEMS; ++k) {
    list_flags(i0 + k, n, iwhere, prev, do_el, fr[k], en[k], lv[k]);
    c[0] += fr[k];
    c[1] += en[k];
    c[2] += lv[k];
  }
  for (int k = 0; k < 3; ++k) sc[k][threadIdx.x] = c[k];
  __syncthreads();
  if (threadIdx.x < 3) {
    int run = 0;
    for (int t = 0; t < BLOCK; ++t) {
      const int v = sc[threadIdx.x][t];
      sc[threadIdx.x][t] = run;
      run += v;
    }
  }
  __syncthreads();
  int64_t pf = (int64_t)tmp[3 * blockIdx.x + 0] + sc[0][threadIdx.x];
  int64_t pe = (int64_t)tmp[3 * blockIdx.x + 1] + sc[1][threadIdx.x];
  int64_t pl = (int64_t)tmp[3 * blockIdx.x + 2] + sc[2][threadIdx.x];
  const int64_t nenter = tmp[3 * nch + 1];
  for (int k = 0; k < LIST_ITEMS; ++k) {
    const int64_t i = i0 + k;
    if (i >= n) break;
    const int32_t var = (int32_t)(i + 1);
    if (fr[k]) {
      index[pf] = var;
      pf++;
    } else {
      const int64_t ar = i - pf;  // actives before i
      index[n - 1 - ar] = var;
    }
    if (en[k]) {
      indx2[nenter - 1 - pe] = var;
      pe++;
    }
    if (lv[k]) {
      indx2[n - 1 - pl] = var;
      pl++;
    }
  }
}
void launch_freev_lists(Queue &q, int64_t n, const iw_t *iwhere, const int8_t *prevfree,
                        int do_enterleave, int32_t *index, int32_t *indx2, int32_t *scan_tmp) {
  const int nch = (int)((n + LIST_CHUNK - 1) / LIST_CHUNK);
  hipLaunchKernelGGL(list_count_kernel, dim3(nch), dim3(BLOCK), 0, q.stream, n, iwhere, prevfree,
                     do_enterleave, scan_tmp);
  hipLaunchKernelGGL(list_scan_kernel, dim3(1), dim3(BLOCK), 0, q.stream, nch, scan_tmp);
  hipLaunchKernelGGL(list_write_kernel, dim3(nch), dim3(BLOCK), 0, q.stream, n, iwhere, prevfree,
                     do_enterleave, scan_tmp, nch, index, indx2);
  q.launches += 3;
}


}  // namespace lbk
