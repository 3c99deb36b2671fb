// k_sort.hip -- ordering of breakpoints: (t, index) radix sorts (rocPRIM), the small bitonic sort of
// freev's changed-row list, the device merge of all-gathered record chunks
// (part of the gfx950 kernel set; kernels_common.hpp has the overview)
#include "kernels_common.hpp"

#include <rocprim/rocprim.hpp>

namespace lbk {

size_t sort_pairs_temp_bytes(size_t count) {
  size_t b1 = 0, b2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b1, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, count, 0, 64,
                                  (hipStream_t)0);
  (void)rocprim::radix_sort_pairs(nullptr, b2, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (const uint64_t *)nullptr, (uint64_t *)nullptr, count, 0, 32,
                                  (hipStream_t)0);
  size_t b3 = 0;
  (void)rocprim::radix_sort_keys(nullptr, b3, (const uint32_t *)nullptr, (uint32_t *)nullptr, count, 0, 32,
                                 (hipStream_t)0);
  b1 = b1 > b2 ? b1 : b2;
  return b1 > b3 ? b1 : b3;
}
// ---- ascending order for a list of <= 2^18 row numbers (freev's changed rows): the list is
//      appended with an atomic counter, i.e. in an order that may change from run to run, and
//      formk's patch sums run over it -- sorted, the sums are reproducible bit for bit ----
constexpr int SMALL_SORT = 2048;
__global__ __launch_bounds__(BLOCK) void sort_u32_small_kernel(uint32_t *keys, uint32_t cnt, int npow2) {
  __shared__ uint32_t sm[SMALL_SORT];
  // (the network is only as large as the list: npow2 = the power of two >= cnt, <= SMALL_SORT)
  for (int k = threadIdx.x; k < npow2; k += BLOCK) sm[k] = (uint32_t)k < cnt ? keys[k] : 0xFFFFFFFFu;
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1)      // bitonic network, one workgroup
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int k = threadIdx.x; k < npow2 / 2; k += BLOCK) {
        const int lo = 2 * k - (k & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint32_t a = sm[lo], b = sm[hi];
        if ((a > b) == up) sm[lo] = b, sm[hi] = a;
      }
      __syncthreads();
    }
  for (int k = threadIdx.x; k < npow2; k += BLOCK)
    if ((uint32_t)k < cnt) keys[k] = sm[k];
}
// the same with the list's length still on the device (freev's position counter, which may exceed what was
// stored): lists of up to SMALL_SORT rows are sorted in place, longer ones are left alone -- the eager patch
// chain of formk (solver_subspace.inl) then reports "not served" and the host takes the ordinary route
__global__ __launch_bounds__(BLOCK) void sort_u32_small_dev_kernel(uint32_t *keys, const uint32_t *cnt_ptr) {
  __shared__ uint32_t sm[SMALL_SORT];
  const uint32_t cnt = *cnt_ptr;
  if (cnt <= 1 || cnt > (uint32_t)SMALL_SORT) return;  // (uniform)
  int npow2 = 2;
  while ((uint32_t)npow2 < cnt) npow2 <<= 1;
  for (int k = threadIdx.x; k < npow2; k += BLOCK) sm[k] = (uint32_t)k < cnt ? keys[k] : 0xFFFFFFFFu;
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int k = threadIdx.x; k < npow2 / 2; k += BLOCK) {
        const int lo = 2 * k - (k & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint32_t a = sm[lo], b = sm[hi];
        if ((a > b) == up) sm[lo] = b, sm[hi] = a;
      }
      __syncthreads();
    }
  for (int k = threadIdx.x; k < npow2; k += BLOCK)
    if ((uint32_t)k < cnt) keys[k] = sm[k];
}
void launch_sort_u32_small_dev(Queue &q, uint32_t *keys, const uint32_t *cnt_ptr) {
  hipLaunchKernelGGL(sort_u32_small_dev_kernel, dim3(1), dim3(BLOCK), 0, q.stream, keys, cnt_ptr);
  LB_LAUNCHED(q);
}
int small_sort_cap() { return SMALL_SORT; }
uint32_t *launch_sort_u32(Queue &q, void *d_temp, size_t temp_bytes, uint32_t *keys, uint32_t *scratch,
                          uint32_t count) {
  if (count <= 1) return keys;
  if (count <= (uint32_t)SMALL_SORT) {
    int npow2 = 2;
    while ((uint32_t)npow2 < count) npow2 <<= 1;
    hipLaunchKernelGGL(sort_u32_small_kernel, dim3(1), dim3(BLOCK), 0, q.stream, keys, count, npow2);
    LB_LAUNCHED(q);
    return keys;
  }
  (void)rocprim::radix_sort_keys(d_temp, temp_bytes, keys, scratch, (size_t)count, 0, 32, q.stream);
  LB_LAUNCHED(q);
  return scratch;
}
// ---- several ranks: the all-gathered record chunks (one sorted run per rank) merged ON THE DEVICE ----
// all = nranks blocks of `stride` doubles: { count, more, records[chunk][recl] }, every block in
// (t, global index) order.  A stable sort on t of the concatenation in rank order IS the (t, global
// index) order of the union (ranks own ascending row blocks).  out = { header[4 nranks] = count, more,
// t and index of the last record of every rank | merged records | one byte per merged record: its rank }.
__global__ __launch_bounds__(BLOCK) void merge_keys_kernel(const double *__restrict__ all, int nranks,
                                                           uint32_t chunk, int recl, size_t stride,
                                                           uint64_t *keys, uint32_t *vals, double *out) {
  const size_t S = (size_t)nranks * chunk;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (size_t)gridDim.x * blockDim.x) {
    const int rk = (int)(s / chunk);
    const uint32_t k = (uint32_t)(s % chunk);
    const double *base = all + (size_t)rk * stride;
    const uint32_t lr = (uint32_t)base[0];
    keys[s] = k < lr ? (uint64_t)__double_as_longlong(base[2 + (size_t)k * recl]) : ~0ull;
    vals[s] = (uint32_t)s;
    if (k == 0) {
      out[4 * rk + 0] = base[0];
      out[4 * rk + 1] = base[1];
      out[4 * rk + 2] = lr ? base[2 + (size_t)(lr - 1) * recl] : 0.0;
      out[4 * rk + 3] = lr ? base[2 + (size_t)(lr - 1) * recl + 1] : 0.0;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void merge_permute_kernel(const double *__restrict__ all, int nranks,
                                                              uint32_t chunk, int recl, size_t stride,
                                                              const uint64_t *__restrict__ keys,
                                                              const uint32_t *__restrict__ vals, double *out) {
  const size_t S = (size_t)nranks * chunk;
  double *recs = out + 4 * (size_t)nranks;
  unsigned char *rb = reinterpret_cast<unsigned char *>(recs + S * (size_t)recl);
  const size_t total = S * (size_t)recl;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t p = e / recl;
    const int f = (int)(e % recl);
    if (keys[p] == ~0ull) continue;  // (slots beyond the records the ranks sent)
    const uint32_t s = vals[p];
    const int rk = (int)(s / chunk);
    const uint32_t k = s % chunk;
    recs[e] = all[(size_t)rk * stride + 2 + (size_t)k * recl + f];
    if (f == 0) rb[p] = (unsigned char)rk;
  }
}
void launch_merge_chunks(Queue &q, int nranks, uint32_t chunk, int recl, size_t stride, const double *all,
                         uint64_t *keys0, uint64_t *keys1, uint32_t *vals0, uint32_t *vals1, void *d_temp,
                         size_t temp_bytes, double *out) {
  const size_t S = (size_t)nranks * chunk;
  hipLaunchKernelGGL(merge_keys_kernel, dim3(grid_for((int64_t)S, 1)), dim3(BLOCK), 0, q.stream, all, nranks,
                     chunk, recl, stride, keys0, vals0, out);
  LB_LAUNCHED(q);
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, keys0, keys1, vals0, vals1, S, 0, 64, q.stream);
  LB_LAUNCHED(q);
  hipLaunchKernelGGL(merge_permute_kernel, dim3(grid_for((int64_t)(S * recl), 4)), dim3(BLOCK), 0, q.stream, all,
                     nranks, chunk, recl, stride, keys1, vals1, out);
  LB_LAUNCHED(q);
}
void launch_sort_by_idx(Queue &q, void *d_temp, size_t temp_bytes, const uint32_t *idx_in,
                        uint32_t *idx_out, const uint64_t *keys_in, uint64_t *keys_out,
                        size_t count) {
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, idx_in, idx_out, keys_in, keys_out, count, 0,
                                  32, q.stream);
  LB_LAUNCHED(q);
}
void launch_sort_pairs(Queue &q, void *d_temp, size_t temp_bytes, const uint64_t *keys_in,
                       uint64_t *keys_out, const uint32_t *idx_in, uint32_t *idx_out,
                       size_t count) {
  (void)rocprim::radix_sort_pairs(d_temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, count, 0,
                                  64, q.stream);
  LB_LAUNCHED(q);
}

}  // namespace lbk
