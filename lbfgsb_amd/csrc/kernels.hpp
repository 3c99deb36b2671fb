// kernels.hpp -- launch interface of the gfx950 kernels (k_*.hip).
//
// Every function enqueues work on `q.stream` and returns immediately; results
// of reductions land in q.d_res (device) after the finalize kernel and are
// fetched by the solver (solver.hip) in one D2H copy per phase.  No function
// here synchronises, allocates or frees (graph-capturable, cdna guide G9).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lbk {

constexpr int BLOCK = 256;        // 4 wave64 per workgroup
constexpr int MAX_BLOCKS = 2048;  // 256 CUs x 8 workgroups, grid-stride beyond that
constexpr int GRAM_BLOCKS = 1024;
constexpr int MAXM = 32;          // LBFGSB_MAX_M
constexpr int RES_MAX = 8 * MAXM + 16;  // >= 6*MC slots of cmprlb_wtv(newrow), 8*MC+15 of update_scan(newrow)
// update pass with formk's new-row sums at col - 1 > 20 (k_update.hip): several launches over a part of the columns
// each; their results land behind the merged layout in d_res (split_base) and are merged into the one-launch layout
// (any col - 1 > 20, k_update.hip "the split pass": parts of <= 16 columns)
constexpr int SPLIT_SLOTS = 8 * 20 + 16, SPLIT_COLS = 16, SPLIT_MAXPARTS = 64;

// per-context launch options (lbfgsb_hip_set_option; nothing is read from the environment)
struct Tune {
  int wgrid = 0;    // workgroups of the passes over W: 0 = what is resident for the kernel launched
                    // (grid_for_w asks the runtime: 256 ... 768), > 0 = this many
  int pipe = -1;    // two trips in flight per wave: -1 / 1 default rule (pipe_on), 0 off
  int pair = 2;     // MC = 20 update pass with new-row sums: lane pairs share the per-column
                    // accumulators; 0 off, 1 one trip in flight, 2 two trips
  int split_from = 20;  // update pass with the new-row sums: split over the columns beyond this many old pairs ...
  int split_cols = 16;  // ... into parts of at most this many (experiment: 10 / 10 = two MC = 10 launches at m = 20)
  int gram_rows = 0;  // formk from scratch: 1 = the LDS-slab kernel instead of the quad kernel
  int pair_cw = 1;    // compact W, MC = 10 update pass with the new-row sums: lane pairs share the column accumulators
};

// one reduction waiting to be finalized: `nblocks` partials per slot in part[slot * pstride + block],
// results to d_res[off + slot] (nsum sums, then nmin minima, then nmax maxima)
struct FinJob {
  const double *part;
  int pstride, nblocks, off, nsum, nmin, nmax;
};

// launch queue + reduction scratch owned by the context
struct Queue {
  hipStream_t stream;
  double *d_part;   // [rows][MAX_BLOCKS] block partials (row = output slot)
  // two small partial-sum matrices (ALT_SLOTS slots) for kernels whose finalize is parked (k_misc.hip,
  // finalize): part_sel = 1 / 2 makes the next launch write there, hold_fin parks its job
  static constexpr int ALT_SLOTS = 8;
  double *d_part_alt[2] = {nullptr, nullptr};
  int part_sel = 0;
  bool hold_fin = false;
  FinJob held[2];
  int nheld = 0;
  double *part() const { return part_sel == 0 ? d_part : d_part_alt[part_sel - 1]; }
  // finalize as publisher (single-rank contexts): device views of the host mirror of d_res and of the
  // sequence word, the workgroup counter, the number of the last publishing launch
  bool fin_publish = false;
  double *hd_pub = nullptr;
  unsigned long long *hd_fin_flag = nullptr;
  unsigned int *d_fin_count = nullptr;
  unsigned long long fin_seq = 0;
  int64_t launches_at_fin = -1;  // q.launches right after the last publishing finalize launch
  double *d_res;    // [RES_MAX or gram size] finalized results, device
  double *d_gpart;  // gram partials [E][GRAM_BLOCKS]
  int64_t launches;
  int res_off = 0;  // finalize writes d_res[res_off + slot] (lets two phases share one fetch)
  // m > 32: the parts of a split update pass never start below this offset of d_res (behind the LONGEST merged
  // layout the context can have + the deferred line-search sums, whatever the current number of pairs)
  int split_base_min = 0;
  bool nt = false;  // nontemporal loads in the W-pass kernels (W much larger than the Infinity Cache)
  Tune tune{};
  // first kernel launch that failed since the last check (hipGetLastError right behind the launch,
  // so that the error is reported with the kernel's name and not at the next stream sync)
  const char *launch_err_where = nullptr;
  hipError_t launch_err = hipSuccess;
  void launched(const char *where) {
    launches++;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && launch_err == hipSuccess) launch_err = e, launch_err_where = where;
  }
};
#define LB_LAUNCHED(q) (q).launched(__func__)

// the circular correction-pair store: Ws, Wy column-major n x m, leading
// dimension ld (multiple of 32 rows).  head is 1-based like the reference.
template <typename T>
struct WStore {
  T *ws;
  T *wy;
  int64_t ld;
  int m;
  const T *zero;  // >= 64 bytes of zeros: what the unroll slots beyond the stored pairs read
  // Tile-local free-row layout ("compact W"): nullptr = every column holds its rows in
  // natural order.  Else one bit per row: inside each aligned tile of CW_TILE rows a column stores the rows whose
  // bit is set first (ascending: the order of the reference's Index(1:nfree), src/lbfgsb.f90:2044-2054), the others
  // behind them.  Only the kernels that take the layout into account may be handed a WStore with lmask set
  // (update_scan, subsm_update, the record gathers and formk's patch); the solver's W() hands out natural order.
  const uint64_t *lmask = nullptr;
};
constexpr int CW_TILE = 128;  // rows per layout tile = what one wave64 covers per trip with two rows per lane

// iwhere (cauchy's per-variable status, -3..3) is kept as one byte per row on the device; the
// reference's int32 layout exists only in export_state / import_state
using iw_t = int8_t;
// nbd as the hot passes read it: one byte per row, a private copy of the caller's int32 array
// (values 0..3; made by nbd_pack_kernel) -- 3 bytes per row less in each pass over W
using nb_t = int8_t;

// small coefficient vectors travel as kernel arguments (scalar loads)
struct Coef {
  double a[2 * MAXM];
};

// A pair that matupd has accepted but that is not stored in W yet (see "pending pair" in
// kernels_common.hpp): logical column col-1 is  y = T(g - r), s = T(stp * d)  until it is committed.
struct Pend {
  int on;      // 0: every column is in W
  double stp;  // step length of the accepted trial
  int impl;    // 1: d itself is not stored either -- the unit first trial step was accepted, so
               //    d = x - t (x the accepted point, t the previous iterate); the `pd` argument of
               //    the kernels then points at t and s = T(x - t)
};

int grid_for(int64_t n, int vec);
// the same for the passes over W: as many workgroups as are resident for `kernel` (1-3 per CU)
int grid_for_w(const Queue &q, int64_t n, int vec, const void *kernel = nullptr);
// compile-time column capacity the kernels are unrolled to for `col` pairs (5, 10, 20, 32).
// Reduction slots that depend on col use MC = maxc_for(col) as their stride.
int maxc_for(int col);

// ---- one-off (START) ----------------------------------------------------
// active (ref :965-1040): clip x, init iwhere.  res: [0]=#projected (sum),
// [1]=#(nbd!=0) (sum), [2]=#(nbd!=2) (sum), [3]=nbdd (sum)
template <typename T>
void launch_active(Queue &q, int64_t n, T *x, const T *l, const T *u, const int32_t *nbd,
                   iw_t *iwhere, int8_t *wasfree);
// errclb (ref :1601-1643): res max-slots: [0]=largest 1-based global index with invalid nbd
// (0 if none), [1]=largest index with l>u and nbd==2; [2], [3], [4] = 1 if some l_i / u_i / nbd_i
// differs from the first one of this rank's rows (uniform-bounds detection, see `ub` below).
template <typename T>
void launch_errclb(Queue &q, int64_t n, int64_t row0, const T *l, const T *u,
                   const int32_t *nbd);

// ---- projgr (ref :2594-2622): res max-slot [0] = sbgnrm -------------------
template <typename T>
void launch_projgr(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                   const int32_t *nbd, const T *g);
// level-1 doors (k_misc.hip): out = a - b; v = alpha v; sum slot 0 = a'b
template <typename T>
void launch_vec_sub(Queue &q, int64_t n, const T *a, const T *b, T *out);
template <typename T>
void launch_vec_scale(Queue &q, int64_t n, double alpha, T *v);
template <typename T>
void launch_dot(Queue &q, int64_t n, const T *a, const T *b);

// ---- W'v (the WS/WY correction-pair matvec) -------------------------------
// res sum-slots [0..col) = Wy' v, [MC..MC+col) = Ws' v (logical column order)
template <typename T>
void launch_wtv(Queue &q, int64_t n, WStore<T> w, int head, int col, const T *v);
template <typename T>
void launch_wtv_nofinalize(Queue &q, int64_t n, WStore<T> w, int head, int col, const T *v);

// ---- cauchy (ref :1157-1532) ----------------------------------------------
// scan (:1270-1330): updates iwhere, writes tbrk (breakpoint t_i > 0; +inf = moves
// without bound; -1 = does not move).  With MC = (col ? maxc_for(col) : 0), res sum-slots:
// [0..col) Wy'd, [MC..MC+col) Ws'd, [2MC] f1, [2MC+1] nbreak, [2MC+2] #moving-without-
// breakpoint, [2MC+3] #of those with g!=0; min-slot [2MC+4] = bkmin (+inf if none)
template <typename T>
void launch_cauchy_scan(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                        const int32_t *nbd, const T *g, iw_t *iwhere, T *tbrk,
                        WStore<T> w, int head, int col);
// candidates with (lo_t, lo_i) < (t, gidx) and t <= hi_t, appended (unordered)
// to keys/idx (capacity cap); *d_count receives the total number found.
template <typename T>
void launch_cauchy_window(Queue &q, int64_t n, int64_t row0, const T *tbrk, double lo_t,
                          int64_t lo_i, double hi_t, uint64_t *keys, uint32_t *idx,
                          uint32_t cap, uint32_t *d_count);
// keys for a full sort of the remaining breakpoints (non candidates -> UINT64_MAX)
template <typename T>
void launch_cauchy_allkeys(Queue &q, int64_t n, int64_t row0, const T *tbrk, double lo_t,
                           int64_t lo_i, uint64_t *keys, uint32_t *idx);
// stable LSD radix sort of (key, idx) pairs, count elements; temp storage query
// when d_temp == nullptr.  Uses both halves of the double buffers; the sorted
// result is left in keys_out/idx_out.
size_t sort_pairs_temp_bytes(size_t count);
void launch_sort_pairs(Queue &q, void *d_temp, size_t temp_bytes, const uint64_t *keys_in,
                       uint64_t *keys_out, const uint32_t *idx_in, uint32_t *idx_out,
                       size_t count);
// ascending order for a list of row numbers (freev's changed rows, <= 2^18): in place for short lists
// (one-workgroup bitonic network), through `scratch` otherwise; returns where the sorted list is
uint32_t *launch_sort_u32(Queue &q, void *d_temp, size_t temp_bytes, uint32_t *keys, uint32_t *scratch,
                          uint32_t count);
// several ranks: merge the all-gathered record chunks (nranks sorted runs, blocks of `stride` doubles:
// count, more, records[chunk][recl]) into ONE (t, global index)-ordered run on the device.
// out = { 4 doubles per rank: count, more, t and index of its last record | the merged records |
// one byte per merged record: the rank it came from }
void launch_merge_chunks(Queue &q, int nranks, uint32_t chunk, int recl, size_t stride, const double *all,
                         uint64_t *keys0, uint64_t *keys1, uint32_t *vals0, uint32_t *vals1, void *d_temp,
                         size_t temp_bytes, double *out);
// stable sort of the same pairs by idx (first pass of a (t, idx) lexicographic order)
void launch_sort_by_idx(Queue &q, void *d_temp, size_t temp_bytes, const uint32_t *idx_in,
                        uint32_t *idx_out, const uint64_t *keys_in, uint64_t *keys_out,
                        size_t count);
// records for `cnt` breakpoints listed in idx (local rows): rec[k*(2col+4)+..] =
// { t, global index, d_i (= -g_i), zibp (= bound - x_i), Wy(i,0..col), Ws(i,0..col) }
// (t comes from keys[k]: the sort key is the bit pattern of the breakpoint time)
template <typename T>
void launch_cauchy_gather(Queue &q, const uint32_t *idx, const uint64_t *keys, uint32_t cnt,
                          int64_t row0, const T *x, const T *l, const T *u, const T *g, WStore<T> w,
                          int head, int col, const T *pr, const T *pd, Pend pe, double *rec);
// one-sync fast path: msg = { *d_count, 0, records of the first min(*d_count, cap) candidates }
template <typename T>
void launch_cauchy_gather_dyn(Queue &q, const uint32_t *idx, const uint64_t *keys,
                              const uint32_t *d_count, uint32_t cap, int64_t row0, const T *x,
                              const T *l, const T *u, const T *g, WStore<T> w, int head, int col,
                              const T *pr, const T *pd, Pend pe, double *msg);
// the window compaction with the breakpoint times recomputed per row (no stored tbrk)
template <typename T>
void launch_cauchy_window_fly(Queue &q, int64_t n, int64_t row0, const T *x, const T *l, const T *u,
                              const nb_t *nbd, const T *g, const iw_t *iwhere, double lo_t,
                              int64_t lo_i, double hi_t, uint64_t *keys, uint32_t *idx, uint32_t cap,
                              uint32_t *d_count, int ub = 0);
// cauchy's iwhere update (:1284-1291) alone
template <typename T>
void launch_iwhere_update(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                          const int32_t *nbd, const T *g, iw_t *iwhere);
// the Cauchy point as a vector from (x, g, l, u, iwhere-after-the-walk, tsum)
template <typename T>
void launch_xcp_fill(Queue &q, int64_t n, const T *x, const T *g, const T *l, const T *u,
                     const iw_t *iwhere, double tsum, T *dst);
// ---- parallel GCP search for col > 0 (LBFGSB_F_PARALLEL_GCP; see k_cauchy.hip) ----
size_t scan_temp_bytes(size_t count);
void launch_scan(Queue &q, void *d_temp, size_t temp_bytes, const double *in, double *out,
                 size_t count, int exclusive);
template <typename T>
void launch_pgcp_gather(Queue &q, const uint32_t *idx, const uint64_t *keys, int64_t nb, int64_t nbp,
                        const T *x, const T *l, const T *u, const T *g, WStore<T> w, int head, int col,
                        double theta, const T *pr, const T *pd, Pend pe, double *tt, double *dd,
                        double *a0, double *wb, double *uu, double *gi, int64_t row0);
void launch_pgcp_mergekeys(Queue &q, int nranks, int64_t nbp, int narr, const double *counts,
                           const double *G, uint64_t *keys, uint32_t *vals);
void launch_pgcp_permute(Queue &q, int64_t NB, int64_t NBp, int64_t nbp, int narr, const uint32_t *vals,
                         const double *G, double *out, const int *map);
size_t f2scan_temp_bytes(size_t count);
void launch_pgcp_f2(Queue &q, void *d_temp, size_t temp_bytes, int64_t nb, double f2_0, double cl,
                    const double *df2, double *maps, double *F2);
template <typename T>
void launch_gcp_rest_mass(Queue &q, int64_t n, const T *g, const T *tbrk, double tstar);
void launch_pgcp_last(Queue &q, int64_t nb, int64_t nbp, int col2, const double *uu, double *uu_last);
void launch_pgcp_dtp(Queue &q, int64_t nb, int64_t nbp, int col2, const double *tt, const double *pp,
                     double *qq);
void launch_pgcp_terms(Queue &q, int64_t nb, int64_t nbp, int col2, double theta, const double *mm,
                       const double *p0, const double *tt, const double *dd, const double *a0,
                       const double *wb, const double *pp, const double *sq, double *df2, double *a1);
void launch_pgcp_f1(Queue &q, int64_t nb, double f2_0, const double *tt, const double *sf2,
                    const double *a1, double *df1);
void launch_pgcp_find(Queue &q, int64_t nb, double f1_0, double f2_0, const double *tt,
                      const double *sf1, const double *sf2);  // res min-slot [0] = k* (or +inf)
void launch_pgcp_pick(Queue &q, int64_t ks, int64_t nb, int64_t nbp, int col2, double f1_0, double f2_0,
                      const double *tt, const double *sf1, const double *sf2, const double *pp,
                      const double *uu_last, const double *sq, const uint32_t *idx, const double *gi,
                      double *out);
// tbrk as a vector from (x, l, u, nbd, g, iwhere-after-the-scan, BEFORE the walk fixes rows)
template <typename T>
void launch_tbrk_fill(Queue &q, int64_t n, const T *x, const T *l, const T *u, const int32_t *nbd,
                      const T *g, const iw_t *iwhere, T *tbrk);
// finish (:1425-1433, :1515): fix every processed breakpoint variable at its bound,
// move the others by tsum*d.  processed = (t, gidx) <= (last_t, last_i).
// count != 0: res[0] = number of rows fixed (finalize launched).
template <typename T>
void launch_cauchy_finish(Queue &q, int64_t n, int64_t row0, const T *x, const T *l, const T *u,
                          const T *g, const T *tbrk, iw_t *iwhere, T *xcp, double tsum,
                          double last_t, int64_t last_i, int count = 0);

// rows fixed by a short walk (list entry = global row * 2 + upper?): iwhere = 1 / 2
void launch_cauchy_fix(Queue &q, const int64_t *list, int count, int64_t row0, int64_t n,
                       iw_t *iwhere);

// ---- freev (ref :1980-2059) -------------------------------------------------
// res sum-slots: [0]=nfree, [1]=nenter, [2]=nleave, [3]=rows whose status changed; updates wasfree.
// chg (optional): rows whose free/active status changed are appended (unordered) as
// row | 0x80000000 if it LEFT the free set ([3] may exceed chg_cap: the list is then truncated).
// cnt2: two zero-initialised position counters, parity = which one this launch uses (it zeroes the other)
// write = 0: counts and list only (speculative pass); launch_freev_apply then updates wasfree from the list
void launch_freev_count(Queue &q, int64_t n, const iw_t *iwhere, int8_t *wasfree,
                        uint32_t *chg, uint32_t chg_cap, uint32_t *cnt2, int parity, int write = 1);
void launch_freev_apply(Queue &q, const uint32_t *chg, const uint32_t *cnt_ptr, uint32_t cap, int8_t *wasfree);
// mirror of Index / Indx2 (1-based global numbers, reference ordering).  prev = wasfree
// BEFORE launch_freev_count of this iteration (copy kept by the solver).
void launch_freev_lists(Queue &q, int64_t n, const iw_t *iwhere, const int8_t *prevfree,
                        int do_enterleave, int32_t *index, int32_t *indx2, int32_t *scan_tmp);

// ---- formk inner products (ref :1756-1851, from scratch) --------------------
// res (in d_res, E = 2col^2+col entries):
//   [ (i*(i+1)/2 + j) ]             i>=j : sum_free Wy_i Wy_j
//   [ T + (i*(i+1)/2 + j) ]         i>=j : sum_act  Ws_i Ws_j       (T = col(col+1)/2)
//   [ 2T + i*col + j ]              all  : sum_{i>j ? act : free} Ws_i Wy_j
template <typename T>
void launch_formk_gram(Queue &q, int64_t n, WStore<T> w, int head, int col,
                       const iw_t *iwhere);

// cmprlb + the first matvec of subsm (W'r, :2742-2754) in one pass over W.
// res sum-slots (MC = maxc_for(col)): [0..col) Wy'r, [MC..MC+col) Ws'r; with newrow also the
// new row/column of formk's WN1 for the pair in logical column col-1 (ref :1756-1793):
// [2MC..) sum_free Wy_new Wy_j, [3MC..) sum_act Ws_new Ws_j, [4MC..) sum_act Ws_new Wy_j,
// [5MC..) sum_free Ws_j Wy_new.  r itself is not stored: launch_subsm_update recomputes it.
// The Cauchy point is evaluated per row from (x, g, iwhere, tsum), see xcp_free in kernels_common.hpp.
template <typename T>
void launch_cmprlb_wtv(Queue &q, int64_t n, const T *x, const T *g, double tsum,
                       const iw_t *iwhere, WStore<T> w, int head, int col, double theta,
                       const Coef &a, int newrow, const T *pr, const T *pd, Pend pe);
// formk patches (ref :1801-1851): signed Gram over the listed rows (+ entered, - left the free
// set) for the first upcl logical columns; res layout as launch_formk_gram with col = upcl.
template <typename T>
void launch_formk_patch(Queue &q, const uint32_t *chg, uint32_t cnt, WStore<T> w, int head, int upcl);
// the same queued right behind freev's counting pass, the list's length still on the device (*cnt_ptr): lists of
// up to `cap` rows (<= small_sort_cap(), sorted by launch_sort_u32_small_dev first) are patched, longer ones
// are not; res = the 2 upcl^2 + upcl sums, then one more: 1.0 if the list was too long
template <typename T>
void launch_formk_patch_dev(Queue &q, const uint32_t *chg, const uint32_t *cnt_ptr, uint32_t cap, WStore<T> w,
                            int head, int upcl);
void launch_sort_u32_small_dev(Queue &q, uint32_t *keys, const uint32_t *cnt_ptr);
int small_sort_cap();

// ---- subsm (ref :2676-2885) --------------------------------------------------
// update (cmprlb's r recomputed, :2770-2816, :2824-2827) + the line-search set-up of mainlb
// :720-722 / lnsrlb :2196-2236 in one pass: zout = projected subspace point from the Cauchy
// point (evaluated per row from x, g, l, u, iwhere, tsum), dvec = zout - x, tvec = x, r = g.
// cf = the coefficients the preceding launch_cmprlb_wtv used (all zero, with tsum = 0, for the
// reference's unconstrained shortcut r = -g, cmprlb :1560-1563: the general formula then gives
// exactly -g); wv = K^-1 W'r.
// res sum-slots: [0] = #bound hits (iword), [1] = dd_p (= g'(z-x)), [2] = dtd ; min-slot [3] =
// stpmx candidate.  xout (= the caller's x, or nullptr): also store the first trial point of the
// line search, x = z, when its step length is known to be 1 (:2265).  pr: the vector a pending
// pair's y is formed from (r of the previous line search); rout / tvec: where r = g and t = x are
// stored -- nullptr with ping-pong iterate buffers, where they are a change of roles, not copies.
template <typename T>
void launch_subsm_update(Queue &q, int64_t n, double tsum, T *zout, const T *pr, T *rout, const T *l,
                         const T *u, const nb_t *nbd, const iw_t *iwhere, const T *xx, const T *gg,
                         WStore<T> w, int head, int col, double theta, const Coef &cf,
                         const Coef &wv, T *dvec, T *tvec, T *xout, int do_stpmx, Pend pe,
                         const T *pd, int ub = 0);
// the Newton direction of the free rows as a vector (0 elsewhere) -- backtracking branch only
template <typename T>
void launch_subsm_dir(Queue &q, int64_t n, const T *xcp, const iw_t *iwhere, const T *xx,
                      const T *gg, WStore<T> w, int head, int col, double theta, const Coef &cf,
                      const Coef &wv, T *ndir);
// backtrack (:2836-2863): res min-slot [0] = alpha; then argmin pass:
// res min-slot [0] = smallest global index attaining alpha (as double)
template <typename T>
void launch_subsm_alpha(Queue &q, int64_t n, const T *xp, const T *r, const T *l, const T *u,
                        const int32_t *nbd, const iw_t *iwhere);
template <typename T>
void launch_subsm_argalpha(Queue &q, int64_t n, int64_t row0, const T *xp, const T *r,
                           const T *l, const T *u, const int32_t *nbd, const iw_t *iwhere,
                           double alpha);
template <typename T>
void launch_subsm_backtrack(Queue &q, int64_t n, int64_t row0, T *z, const T *xp, T *r,
                            const T *l, const T *u, const iw_t *iwhere, double alpha,
                            int64_t ibd);

// ---- lnsrlb (ref :2174-2275) + mainlb d=z-x (:720-722) ------------------------
// begin: d = z - x, t = x, r = g.  res sum [0]=dtd, [1]=gd ; min [2]=stpmx candidate
template <typename T>
void launch_lnsrlb_begin(Queue &q, int64_t n, const T *z, const T *x, const T *g, const T *l,
                         const T *u, const int32_t *nbd, T *d, T *t, T *r, int do_stpmx);
// trial point (:2264-2270): x = z (stp == 1) or stp*d + t
template <typename T>
void launch_lnsrlb_step(Queue &q, int64_t n, T *x, const T *z, const T *d, const T *t,
                        double stp);
// after f,g evaluation: res sum [0] = g.d ; max [1] = |proj g|_inf (speculative projgr)
template <typename T>
void launch_lnsrlb_eval(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                        const int32_t *nbd, const T *g, const T *d);

// ---- mainlb :812-824 + matupd (ref :2291-2346) --------------------------------
// y = g - r, s = stp*d stored into column itail (1-based physical); with
// MC = maxc_for(col-1), res sum: [0..col-1) = s'Wy_j, [MC..MC+col-1) = Ws_j's for the
// col-1 older columns (logical order from head), [2MC] = y'y
template <typename T>
void launch_update_pairs(Queue &q, int64_t n, const T *g, const T *r, const T *d, double stp,
                         WStore<T> w, int head, int col, int itail);

// the two above fused (one pass over the old columns); slots, MC = maxc_for(col-1):
// [0,MC) s'Wy_j | [MC,2MC) Ws_j's | [2MC] y'y | [2MC+1,3MC+1) Wy_j'd | [3MC+1] y'd |
// [3MC+2,4MC+2) Ws_j'd | [4MC+2] s'd | [4MC+3] f1, nbreak, nunb, nunbnz | [4MC+7] g'd |
// [4MC+8] #iwhere changes | min [4MC+9] bkmin | max [4MC+10] |proj g|
template <typename T>
void launch_update_scan(Queue &q, int64_t n, const T *x, const T *l, const T *u,
                        const nb_t *nbd, const T *g, const T *r, const T *d, int dimpl,
                        double stp, iw_t *iwhere, T *tbrk, WStore<T> w, int head, int col, int itail,
                        int store_pair, int store_iw, int newrow = 0, double cand_hi = -1.0,
                        uint64_t *ckeys = nullptr, uint32_t *cidx = nullptr, uint32_t ccap = 0,
                        uint32_t *ccount = nullptr, int ub = 0);
// ub (launch_update_scan, launch_subsm_update, launch_cauchy_window_fly): UNIFORM BOUNDS.  bit 0: every
// l_i is one value, bit 1: every u_i, bit 2: every nbd_i -- the corresponding pointer then is a
// 64-byte device buffer filled with that value, which every lane reads from its start (a cache hit
// instead of an HBM stream of 8 / 8 / 1 bytes per row).  Detected at START (errclb's pass).
// bit 3 (UB_DICT, with bits 0 and 1 set, bit 2 clear): FEW-VALUED bounds -- the l / u buffers hold tables of
// <= 8 values each and the nbd byte stream carries nbd | l-index << 2 | u-index << 5 (kernels_common.hpp).
// cand_hi >= 0: rows whose breakpoint t lies in [0, cand_hi] are appended (unordered) to
// ckeys / cidx (capacity ccap), *ccount = how many there are (zeroed by the launch)
// newrow (col - 1 <= 10): 4 MC + 4 more sum slots in front of the min / max slots -- the new
// row/column of formk's WN1 with the PRE-walk free set (layout: update_scan_kernel in
// k_update.hip); min slot = 4 MC + 9 + (newrow ? 4 MC + 4 : 0), max slot behind it
inline int update_scan_extra(int nold, int newrow);
// Ws/Wy slot of logical column col-1 <- the pending pair (paths without a subspace pass)
void launch_nbd_pack(Queue &q, int64_t n, const int32_t *nbd, nb_t *out);
// Dictionary-coded bounds (ub bit 3 = UB_DICT, kernels_common.hpp): bound arrays with <= 8 distinct values each.
// The tables travel by value; entry j of l / u beyond nl / nu repeats entry 0.
constexpr int UB_DICT = 8;  // the dictionary bit of `ub`
struct BoundTables {
  double l[8], u[8];  // values of the context's real kind, widened exactly
  int nl, nu, nb0;    // entries in use; nb0: the value of a uniform nbd array
};
// res: sum [0] #l_i outside the table, [1] #u_i outside | min [2] smallest such l_i, [3] smallest such u_i
template <typename T>
void launch_dict_probe(Queue &q, int64_t n, const T *l, const T *u, const BoundTables &tb);
// out[i] = nbd_i | index of l_i << 2 | index of u_i << 5
template <typename T>
void launch_nbd_pack_dict(Queue &q, int64_t n, const int32_t *nbd, const T *l, const T *u, const BoundTables &tb,
                          nb_t *out);
// the caller's l, u, nbd against the snapshot the passes over W read (packed byte / constants / tables), bit for
// bit.  res: sum [0] = rows that differ
template <typename T>
void launch_bounds_verify(Queue &q, int64_t n, const T *l, const T *u, const int32_t *nbd, const nb_t *code,
                          int ub, const BoundTables &tb);
// two copies of (l, u, nbd) against each other, bit for bit.  res: sum [0] = rows that differ
template <typename T>
void launch_bounds_same(Queue &q, int64_t n, const T *l0, const T *u0, const int32_t *nb0, const T *l1,
                        const T *u1, const int32_t *nb1);
// d = x - t, z = x: the vectors a lean subsm_update_kernel pass left implicit (Pend::impl)
template <typename T>
void launch_dz_materialise(Queue &q, int64_t n, const T *x, const T *t, T *d, T *z, T *xnew = nullptr,
                           double stp = 1.0);
template <typename T>
void launch_pair_commit(Queue &q, int64_t n, const T *g, const T *r, const T *d, Pend pe,
                        WStore<T> w, int head, int col);

// ---- tile-local free-row layout of W (k_layout.hip) ----
// Re-sort every live column (logical 0 .. col-1 from head) of Ws and Wy, tile by tile, from the layout `lmask`
// describes to the one where the rows with iwhere <= 0 come first (iwhere == nullptr: natural order, every bit
// set), and write the new bits.  Tiles whose bits do not change are skipped.  fp64 contexts only use it.
template <typename T>
void launch_w_relayout(Queue &q, int64_t n, const iw_t *iwhere, uint64_t *lmask, WStore<T> w, int head, int col);
// every bit of rows [0, n) set, the rest clear (natural order)
void launch_lmask_ones(Queue &q, int64_t n, uint64_t *lmask);

// ---- m > 32: unfused tile primitives (k_wide.hip) -------------------------------------------------
// out_i (+)= sum_j (Wy(i,j) a_j) / div + Ws(i,j) b_j over the tc <= 32 logical columns from `head`
// (a = cf.a[0..), b = cf.a[MAXM..)); masked: only rows with iwhere <= 0
template <typename T>
void launch_tile_axpy(Queue &q, int64_t n, WStore<T> w, int head, int tc, const Coef &cf, double div,
                      const iw_t *iwhere, int masked, T *out);
// The r pass of an m > 32 iteration (solver_wide.inl, wide_subspace): the same tiles, with what cmprlb and subsm do
// around them folded into the first and the last one.  first: the sum of a row starts from r0 = -theta (xcp - x) - g
// (cmprlb :1565-1574; plain: -g) on the free rows instead of from `out` -- the Cauchy point evaluated per row
// (xcp_row), neither xcp nor r0 exists as a vector.  last: the finished sum is the Newton direction; the projected
// step (:2780-2816), dd_p (:2824-2827), d = z - x, dtd, the stpmx ratios (:2196-2225) and the stores of
// subsm_update_kernel (trial x / z, d, t, r: any of them may be nullptr) follow in the same kernel.
// res (last only): sum [0] = #bound hits, [1] = dd_p = g'd, [2] = dtd ; min [3] = stpmx
template <typename T>
struct WideTail {
  const T *x, *g, *l, *u;
  const int32_t *nbd;
  double tsum, theta;
  int plain, do_stpmx;
  T *zout, *dvec, *tvec, *rout, *xout;
};
template <typename T>
void launch_tile_axpy_fused(Queue &q, int64_t n, WStore<T> w, int head, int tc, const Coef &cf, const iw_t *iwhere,
                            T *out, int first, int last, const WideTail<T> &wt);
// The same pass as ONE launch over all col <= WIDE_MAXC columns (the coefficients are a kernel argument): no partial
// sum of r goes through memory between tiles, and the pair matupd left pending is read from the vectors it was
// formed from and stored into its W slot here, as subsm_update_kernel does for m <= 32 (cwy / cws = that slot;
// pe.on == 0: every column is in W).  wt.l / wt.u / nbd8 with the uniform / dictionary-coded bounds of `ub`.
constexpr int WIDE_MAXC = 96;
struct CoefWide {
  double a[2 * WIDE_MAXC];  // a = [0, WIDE_MAXC), b = [WIDE_MAXC, ...)
};
template <typename T>
void launch_wide_r_pass(Queue &q, int64_t n, WStore<T> w, int head, int col, const CoefWide &cf, const iw_t *iwhere,
                        const nb_t *nbd8, int ub, const WideTail<T> &wt, Pend pe, const T *pr, const T *pd);
// out = src on the free (want_free) / active rows, 0 elsewhere
template <typename T>
void launch_masked_copy(Queue &q, int64_t n, const T *src, const iw_t *iwhere, int want_free, T *out);
template <typename T>
void launch_rows_gather(Queue &q, const uint32_t *chg, uint32_t cnt, WStore<T> w, int head, int upcl, double *out);
// cauchy's d as a vector from tbrk (moving rows: -g, others 0)
template <typename T>
void launch_cauchy_dvec(Queue &q, int64_t n, const T *g, const T *tbrk, T *out);
// cmprlb's r before the W terms (0 on the rows that are not free)
template <typename T>
void launch_cmprlb_init(Queue &q, int64_t n, const T *x, const T *g, const T *z, const iw_t *iwhere,
                        double theta, int plain, T *out);
// subsm's projected step from the Newton direction vector; res sum [0] = iword count, [1] = dd_p
template <typename T>
void launch_subsm_project(Queue &q, int64_t n, T *z, T *dir, const T *x, const T *g, const T *l, const T *u,
                          const int32_t *nbd, const iw_t *iwhere, double rtheta);

// ---- built-in objectives -------------------------------------------------------
// res sum [0] = f contribution of this rank
template <typename T>
void launch_obj_quadratic(Queue &q, int64_t n, int64_t row0, const T *x, T *g);
template <typename T>
void launch_obj_rosenbrock(Queue &q, int64_t n, int64_t row0, int64_t nglob, const T *x, T *g,
                           double xl, double xr);
template <typename T>
void launch_halo_pack(Queue &q, int64_t n, const T *x, double *out);
// res sum [0] = *d_val (this rank's part of an objective value the caller computed on the device)
void launch_scalar_partial(Queue &q, const double *d_val);

// the stride of the update pass's result layout: the column capacity of the kernel that ran, or -- beyond 32
// columns, where the pass is split into sub-launches and merged -- the next multiple of 32
inline int maxc_stride(int col) { return col <= MAXM ? maxc_for(col) : (col + 31) / 32 * 32; }
inline int update_scan_extra(int nold, int newrow) { return newrow ? 4 * maxc_stride(nold) + 4 : 0; }
// where the sub-launches of a split update pass put their results in d_res: behind the merged layout at `dst`
inline int split_base(int nold, int dst) { return std::max(RES_MAX + 16, dst + 8 * maxc_stride(nold) + 32); }
inline int split_parts(int nold, int cols = SPLIT_COLS) { return (nold + cols - 1) / cols; }
// d_res doubles a context with m pairs needs for a split update pass (parts of >= 5 columns)
inline size_t split_res_len(int m) {
  return m <= 5 ? 0 : (size_t)split_base(m, 1) + (size_t)split_parts(m, 5) * SPLIT_SLOTS + 8;
}

// finalize: partials -> d_res (nsum sums, then nmin mins, then nmax maxes); takes parked jobs along
void launch_finalize(Queue &q, int nblocks, int nsum, int nmin, int nmax);
void finalize_flush(Queue &q);  // launch the parked jobs alone (a fetch that no finalize precedes)
// publish: count doubles from src (device) to dst_host (device address of mapped host memory), then
// *flag_host = seq with release semantics at system scope -- what the host polls instead of waiting
// for the stream (solver.hip, fetch)
void launch_publish(Queue &q, const double *src, double *dst_host, int count, unsigned long long seq,
                    unsigned long long *flag_host);

}  // namespace lbk
