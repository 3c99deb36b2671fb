/*
 * lbfgsb_hip_debug.h -- measurement and test instruments of liblbfgsb_hip.so.
 *
 * Not part of the drop-in surface (include/lbfgsb_hip.h): a caller that replaces the reference's setulb
 * needs nothing from this file.  bench.py, profiles/scripts/ and the tests use these doors to time single
 * kernels on a context's state, to read the in-run hipEvent clocks of the passes over W, and to count what
 * an iteration did (launches, host syncs, routes taken).  Same conventions as lbfgsb_hip.h: plain C,
 * every entry returns LBFGSB_OK or an error code, counters are per context.
 */
#ifndef LBFGSB_HIP_DEBUG_H
#define LBFGSB_HIP_DEBUG_H

#include "lbfgsb_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bare streaming kernel launches, and the same bracketed by hipEvents on the
 * context's stream: *h_ms_per_launch = average duration of `reps` launches. */
int lbfgsb_hip_wtv_launch_only(lbfgsb_hip_ctx *ctx, const void *v, int col, int head);
int lbfgsb_hip_wtv_time(lbfgsb_hip_ctx *ctx, const void *v, int col, int head, int reps,
                        double *h_ms_per_launch);
/* the same for the kernels that carry the matvec inside the iteration: which = 0
 * cmprlb_wtv_kernel (r of cmprlb + W'r of subsm, src/lbfgsb.f90:1565-1583 + :2742-2754),
 * which = 2 the same with formk's new row/column sums riding along (:1756-1793; the variant
 * the iteration runs after a BFGS update), which = 1 the from-scratch formk Gram kernel
 * (:1756-1851), which = 3 subsm_update_kernel (Newton direction + projected step + line-search
 * set-up, :2770-2827, with a pending pair committed to W), which = 4 update_scan_kernel run as
 * the evaluation of a trial point (matupd's and the next cauchy scan's sums, :2335-2336 +
 * :1270-1330; reduces only).  Uses the context's current W and iwhere and, for 3 and 4, the
 * l, u, nbd of the last setulb call; x, g are device pointers; z, r, d, t and the newest
 * column of W are overwritten by which = 3. */
int lbfgsb_hip_kernel_time(lbfgsb_hip_ctx *ctx, int which, const void *x, const void *g, int col,
                           int head, int reps, double *h_ms_per_launch);

/* counters for bench/profiling: kernel launches, host syncs, full breakpoint sorts so far,
 * and the seconds the host spent blocked waiting for the stream */
int lbfgsb_hip_stats(lbfgsb_hip_ctx *ctx, int64_t *launches, int64_t *syncs,
                     int64_t *cauchy_fullsorts, double *wait_seconds);

/* iterations so far whose freev (src/lbfgsb.f90:1980-2059) needed neither its counting pass nor a host
 * sync: no iwhere entry had changed since the previous freev (the update pass counts the entries it
 * changes, the walk knows the rows it fixes), so nobody entered or left the free set */
int lbfgsb_hip_freev_skipped(lbfgsb_hip_ctx *ctx, int64_t *count);

/* several ranks: collectives issued so far (all-gathers of partial sums, of breakpoint records, of
 * halo values) and the bytes THIS rank contributed to them */
int lbfgsb_hip_comm_stats(lbfgsb_hip_ctx *ctx, int64_t *collectives, int64_t *bytes_contributed);

/* how many subspace minimisations so far took the two-pass route (W'Z r in closed form, no
 * cmprlb pass over W: col <= 20, walk of <= 2^20 segments, no stored s_i with its free part a
 * tiny remainder of the column) and
 * how many the three-pass route (cmprlb_wtv_kernel); handed_windows = Cauchy walks whose
 * breakpoints came with the update pass itself (no window pass, no host sync of their own) */
int lbfgsb_hip_path_counts(lbfgsb_hip_ctx *ctx, int64_t *closed_form, int64_t *three_pass,
                           int64_t *handed_windows);

/* number of setulb calls so far whose Cauchy walk ended inside a group of equal breakpoints
 * (see LBFGSB_F_EXACT_TIES) */
int lbfgsb_hip_tie_splits(lbfgsb_hip_ctx *ctx, int64_t *count);

/* host seconds (and how many stretches) between the landing of a trial point's sums and the launch of the
 * storing pass that follows in the same call -- dcsrch, matupd, formt, the Cauchy walk's host part, formk's
 * assembly and factorisations, W'Z r in closed form: the part of an iteration during which the device waits
 * for the host (window / freev syncs inside the stretch included) */
int lbfgsb_hip_host_gap(lbfgsb_hip_ctx *ctx, double *seconds, int64_t *count);
/* Option "compact_w" (lbfgsb_hip_set_option): how often the tiles of W were re-sorted to the free-rows-first layout
 * (packs) and back to natural row order because a kernel that does not know the layout had to run (unpacks);
 * packed = the columns are not in natural order right now; eligible = this context runs its two passes over W on
 * the layout (fp64, m <= 10, no LBFGSB_F_MIRROR_INDEX, option set). */
int lbfgsb_hip_compact_stats(lbfgsb_hip_ctx *ctx, int64_t *packs, int64_t *unpacks, int32_t *packed,
                             int32_t *eligible);
/* ONE host sync of the iteration timed by itself, `reps` times (microseconds: median and minimum): the
 * 8 min(m, 32) + 15 fp64 partials of the widest phase through the library's own fetch -- with a communicator the
 * all-gather over the ranks on the solver's stream (the sums of src/lbfgsb.f90:813-816, 2196-2244
 * completed over the row blocks), the copy into mapped host memory and the poll; without one the publish + poll.
 * A collective: every rank calls it, between runs (not while sums of a run are deferred).  What a --gpus N bench
 * line reports as collective_us. */
int lbfgsb_hip_collective_time(lbfgsb_hip_ctx *ctx, int reps, double *median_us, double *min_us);
/* the same stretches cut at their milestones, accumulated seconds: [0] line search + return to the caller,
 * [1] the caller between the NEW_X return and the re-entry, [2] termination tests + matupd + formt,
 * [3] cauchy (host walk, window syncs if any) + freev, [4] formk's assembly / factorisations, W'Z r, the
 * triangular solves */
int lbfgsb_hip_host_segments(lbfgsb_hip_ctx *ctx, double *seconds5);

/* LBFGSB_F_DEFER_LNSRCH: line-search set-ups whose sums travelled with the next call's fetch, and how many
 * of those had to re-issue their 'FG_LNSRCH' request (backtracking step, ascent direction) */
int lbfgsb_hip_defer_stats(lbfgsb_hip_ctx *ctx, int64_t *deferred, int64_t *reissued);

/* Skipped BFGS updates (src/lbfgsb.f90:822-830) whose next cauchy n-loop (:1270-1330) was taken from the
 * pass that had evaluated the accepted point instead of a scan of its own over all of W (option
 * "skip_reuse", default 1) */
int lbfgsb_hip_skip_stats(lbfgsb_hip_ctx *ctx, int64_t *scans_reused);

/* In-run clocks of the three passes over W of an iteration (hipEvents on the context's stream
 * around every launch, read back at the next host sync): [0] cmprlb_wtv_kernel, [1]
 * update_scan_kernel, [2] subsm_update_kernel -- each reading covers the kernel and the few-us
 * finalize_kernel launched with it.  enable = 1: reset and start; 0: stop; -1: just read.
 * ms_total[3] / count[3] receive the accumulated milliseconds and the number of launches. */
int lbfgsb_hip_pass_clock(lbfgsb_hip_ctx *ctx, int enable, double *ms_total, int64_t *count);

/* memory refreshes so far in the current run (info /= 0 branches of mainlb, src/lbfgsb.f90:620-635, 666-682,
 * 694-710, 757-769, 851-865: the L-BFGS memory is dropped and the iteration restarts with col = 0 -- the
 * calls whose Cauchy walk is long again); reset by task 'START' */
int lbfgsb_hip_refresh_count(lbfgsb_hip_ctx *ctx, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif
