/*
 * lbfgsb_hip.h -- C ABI of the MI355X-native L-BFGS-B inner iteration.
 *
 * This is the drop-in boundary for the hot path of jacobwilliams/lbfgsb: the
 * n-dimensional work inside `mainlb` (reference src/lbfgsb.f90:312-949) runs as
 * hand-written HIP kernels on gfx950; the 2m x 2m algebra (bmv, formt, dpofa,
 * dtrsl, dcsrch) runs on the host inside this library.  The reverse-
 * communication protocol, the `task` strings and the documented isave/dsave/
 * lsave slots are the reference's (src/lbfgsb.f90:88-244).
 *
 * Plain C: pointers and sizes only, no torch/HIP types in the signatures
 * (a hipStream_t travels as void*).  There is NO CPU fallback: every entry
 * point that computes fails with LBFGSB_E_NOGPU when no gfx950 device is
 * usable.
 *
 * The Fortran module `lbfgsb_module` in lbfgsb_amd/fortran/ binds these with
 * iso_c_binding so that reference test/driver1.f90, driver2.f90, driver3.f90
 * link unchanged.
 */
#ifndef LBFGSB_HIP_H
#define LBFGSB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lbfgsb_hip_ctx lbfgsb_hip_ctx; /* opaque */

/* status codes */
enum {
  LBFGSB_OK = 0,
  LBFGSB_E_NOGPU = -100,   /* no usable HIP device / kernel launch failed   */
  LBFGSB_E_ARG = -101,     /* bad argument (n<=0, m<=0, m>LBFGSB_MAX_M ...) */
  LBFGSB_E_ALLOC = -102,   /* device allocation failed                      */
  LBFGSB_E_COMM = -103,    /* RCCL / host reducer failure                   */
  LBFGSB_E_STATE = -104    /* call sequence violates the task protocol      */
};

/* m: the fused passes hold all 2 m operands of a row group in registers and are unrolled for at most
 * LBFGSB_FUSED_M pairs (two passes over W per iteration; beyond 21 stored pairs the first one runs as
 * several launches over a part of the columns each); beyond that a context runs that first pass the same
 * way in front of unfused tile kernels for the subspace steps (k_wide.hip) -- two passes over W as well,
 * a few more vector kernels.  LBFGSB_MAX_M only bounds the host's O(m^2) arrays. */
#define LBFGSB_FUSED_M 32
#define LBFGSB_MAX_M 1024

/* flags for lbfgsb_hip_create */
enum {
  LBFGSB_F_REAL32 = 1,       /* REAL32 build of the reference (lbfgsb_kinds_module.F90:29):
                                fp32 storage + kernels, fp64 partials and host algebra */
  LBFGSB_F_MIRROR_INDEX = 2, /* keep the reference's Index/Indx2 lists (freev,
                                src/lbfgsb.f90:2044-2054, 2014-2035) on the device so
                                that lbfgsb_hip_export_state reproduces `iwa` */
  LBFGSB_F_NO_RETURN_SYNC = 4, /* the caller evaluates f,g on the SAME stream: do not block the
                                host when returning task='FG_LNSRCH' (x is ordered by the stream) */
  LBFGSB_F_PARALLEL_GCP = 8, /* OPT-IN deviation from the reference's arithmetic: when no pair is
                                stored (col = 0: first iteration, after every memory refresh -- the
                                calls where nseg ~ n) the model Hessian is theta*I, the derivative
                                along the projected path is -(1 - theta t) * sum_{t_j >= t} d_j^2,
                                and the generalized Cauchy point is t* = 1/theta in closed form: one
                                elementwise kernel instead of the ordered walk of
                                src/lbfgsb.f90:1378-1497.  Equal to the reference in exact arithmetic;
                                in floating point the reference's f1/f2 recurrence carries its own
                                rounding noise, so tsum (and nseg by a few units) may differ.  The
                                clamp f2 >= epsmch*f2_org (:1483) acts only once the gradient mass
                                still moving is below epsmch of the total: that is checked first and
                                such calls take the exact walk.
                                With pairs stored (col > 0) the flag replaces LONG walks (more than
                                32768 breakpoints within reach) by a full sort + prefix scans of the
                                walk's state on the device, the clamp included (a scan over the maps
                                x -> max(B, x + A)) -- again the reference's result in exact
                                arithmetic.  With several ranks the records of all breakpoints are
                                all-gathered and every rank runs the same (bitwise-reproducible)
                                scans.  Short walks always replay the walk exactly. */
  LBFGSB_F_EXACT_TIES = 16,  /* accepted and ignored: this IS the default behaviour now (round 3).
                                Breakpoints with EQUAL t reach the walk in variable order; the reference
                                pops them in the order of hpsolb's heap (src/lbfgsb.f90:2079, used at
                                :1384-1403).  Sums over a whole group of equal breakpoints are merely
                                reassociated; the two orders differ in effect only when the walk ends
                                INSIDE such a group -- then the order decides which of its variables are
                                fixed at their bounds.  Such calls are detected, counted
                                (lbfgsb_hip_tie_splits) and REPLAYED from the start of the walk in the
                                reference's own order: all breakpoint times travel to the host of
                                every rank (O(n) bytes + an O(n) heap build; every rank pops the same
                                replicated heap and gathers the records of the rows it owns), so the
                                active set equals the reference's bit for bit.  Only the calls that
                                need it pay for it. */
  LBFGSB_F_INDEX_TIES = 32,  /* OPT-OUT of that replay: a walk that ends inside a group of equal
                                breakpoints fixes the group's members in variable order (which members --
                                and, when their rows of W differ, how many -- may then differ from the
                                reference).  For callers who prefer the
                                O(window) cost bound over the reference's tie order: the replay costs
                                what the reference's own walk costs (heap pops over all breakpoints). */
  LBFGSB_F_DEFER_LNSRCH = 64 /* for callers that evaluate f,g on the context's own stream and do nothing
                                else with the context between an 'FG_LNSRCH' return and the re-entry with
                                that evaluation (lbfgsb_hip_minimize with a built-in objective, bench.py).
                                The pass that forms the subspace step also stores the first trial point
                                of the line search, x = z (lnsrlb, src/lbfgsb.f90:2265: the unit step), and
                                produces the numbers its set-up needs (d'd, g'd, stpmx, subsm's iword,
                                :2196-2244, :2820-2828).  By default the call waits for them before it
                                returns 'FG_LNSRCH'.  With this flag it returns at once -- no host sync
                                in that call, implies LBFGSB_F_NO_RETURN_SYNC -- and the numbers come
                                over with the first fetch of the NEXT call, which then runs the set-up
                                and, in the same call, the step of dcsrch that consumes the evaluation:
                                one host round trip per iteration less.  Same arithmetic in the same
                                order: every 'NEW_X' return (x, f, g, isave, dsave, the exported state)
                                is bit for bit the default's.  What differs:
                                 * at such an 'FG_LNSRCH' return dsave / isave / csave do not describe
                                   the line search yet, and lbfgsb_hip_export_state is refused
                                   (LBFGSB_E_STATE) until the next 'NEW_X';
                                 * when the set-up turns out to ask for ANOTHER point than x = z --
                                   subsm's backtracking step (:2830-2879), an ascent direction (:2247)
                                   -- the evaluation the caller has just delivered is dropped (it is not
                                   counted in nfgv) and the call returns 'FG_LNSRCH' once more, with
                                   the point the reference would have asked for (lbfgsb_hip_defer_stats
                                   counts these).
                                Not with LBFGSB_F_MIRROR_INDEX, LBFGSB_F_PARALLEL_GCP or iprint >= 99: such
                                contexts wait as before (m > LBFGSB_FUSED_M: deferred while col <= 96 and the
                                options "wide_tail" / "wide_one" are on, the defaults). */
};

/* -------------------------------------------------------------------------
 * Context.  Replaces the partition of caller memory done by setulb
 * (src/lbfgsb.f90:250-265): Ws, Wy (n x m each, column-major, leading
 * dimension padded to 32 rows), z, r, d, t, xp, iwhere and scratch live in
 * HBM for the life of the context.
 *   n_local   rows owned by this rank          n_global  total rows
 *   row0      global 0-based index of local row 0 (contiguous block sharding)
 * For one GPU: n_local = n_global = n, row0 = 0.
 * stream: a hipStream_t (may be NULL = a private stream is created).
 * ------------------------------------------------------------------------- */
int lbfgsb_hip_create(int64_t n_local, int64_t n_global, int64_t row0, int m, int flags,
                      int device, void *stream, lbfgsb_hip_ctx **out);
void lbfgsb_hip_destroy(lbfgsb_hip_ctx *ctx);
const char *lbfgsb_hip_last_error(void);

/* -------------------------------------------------------------------------
 * Multi-GPU: every n-length reduction of the path is completed across ranks.
 * (a) RCCL on the context's stream (librccl is dlopen'ed on first use):
 *       id = 128-byte ncclUniqueId made by lbfgsb_hip_rccl_unique_id on rank 0
 *       and distributed by the caller (e.g. torch.distributed broadcast).
 *       ONE collective per host sync: the <= 8m+15 fp64 partials of a phase (sums,
 *       minima, maxima together) are all-gathered and reduced on every rank's host in
 *       rank order -- an all-reduce whose result is bit-identical on every rank and
 *       independent of the collective algorithm RCCL picks.
 * (b) a host callback, for launchers that already own a communicator
 *     (MPI, gloo): called with the rank-local partials in host memory; must
 *     return with buf[0..nsum) summed, buf[nsum..nsum+nmin) min-reduced and
 *     the following nmax entries max-reduced over all ranks.  gather: if
 *     non-NULL, must all-gather `bytes` bytes from every rank into out
 *     (rank-major).
 * ------------------------------------------------------------------------- */
int lbfgsb_hip_rccl_unique_id(void *id128);
int lbfgsb_hip_comm_init_rccl(lbfgsb_hip_ctx *ctx, const void *id128, int rank, int nranks);
/* (ncclCommInitRank waits for every rank; the entry gives it LBFGSB_COMM_INIT_TIMEOUT_S seconds -- an
 *  environment variable, default 120; anything but a positive number is LBFGSB_E_ARG -- and returns
 *  LBFGSB_E_COMM after that instead of hanging.  A helper thread is then still inside the call: the process
 *  MUST EXIT after this error (non-zero; do not re-launch in place, do not destroy the context and carry on).
 *  Should the peers arrive later after all, the helper destroys the late communicator itself.)
 * What the context's communicator is: *kind = 0 none (one rank), 1 RCCL, 2 host callbacks; for RCCL
 * *nranks / *rank are the communicator's OWN answers (ncclCommCount, ncclCommUserRank), so a scaling
 * run can show that N ranks really took part. */
int lbfgsb_hip_comm_info(lbfgsb_hip_ctx *ctx, int32_t *nranks, int32_t *rank, int32_t *kind);
typedef int (*lbfgsb_allreduce_fn)(void *user, double *buf, int nsum, int nmin, int nmax);
typedef int (*lbfgsb_allgather_fn)(void *user, const void *in, void *out, int64_t bytes);
int lbfgsb_hip_comm_init_host(lbfgsb_hip_ctx *ctx, lbfgsb_allreduce_fn ar,
                              lbfgsb_allgather_fn ag, void *user, int rank, int nranks);

/* -------------------------------------------------------------------------
 * setulb, device-pointer form.  Mirrors reference setulb (src/lbfgsb.f90:88)
 * argument for argument except that
 *   - n, m, wa, iwa are replaced by the context,
 *   - x, l, u, nbd, g are DEVICE pointers to this rank's rows (real = double,
 *     or float with LBFGSB_F_REAL32; nbd int32), 16-byte aligned,
 *   - f, factr, pgtol are double; f is the GLOBAL objective value,
 *   - task/csave are 60 bytes, blank padded, no terminator (Fortran layout),
 *   - lsave[4] are int32 0/1, isave[44] int32, dsave[29] double, with the
 *     reference's meaning slot for slot (src/lbfgsb.f90:188-242).
 * The caller evaluates f,g on the device whenever task(1:2)=='FG'.
 * l, u and nbd MUST NOT CHANGE between task='START' and the end of the run.  The reference re-reads the three
 * arrays on every call (src/lbfgsb.f90:1270-1330, 2594-2622, 2789-2816); this library's passes over W read a
 * snapshot: a packed one-byte copy of nbd (refreshed on START, after import_state and when the nbd POINTER
 * changes), and for bound arrays found at START to hold one value each, or a few values (<= 8 each), that
 * constant / a table entry selected by the packed byte (lbfgsb_hip_uniform_bounds below).  An edit IN PLACE is
 * detected, not ignored: after the first iteration and then every 32nd (option "bounds_check") the caller's
 * arrays are compared with the snapshot bit for bit, and a difference ends the run with
 * task = 'ERROR: BOUNDS CHANGED DURING RUN' (isave(35), info, = -10).  Passing OTHER array pointers than at
 * START is allowed: the constants / tables are dropped and the arrays are streamed from then on.  A caller that
 * has to change bounds starts a new run (task='START').
 * ------------------------------------------------------------------------- */
int lbfgsb_hip_setulb_dev(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u,
                          const int32_t *nbd, double *f, void *g, double factr, double pgtol,
                          char *task, int iprint, char *csave, int32_t *lsave, int32_t *isave,
                          double *dsave);

/* -------------------------------------------------------------------------
 * setulb, device-pointer form with PING-PONG iterate buffers.  The reference's line search starts
 * every iteration with two n-vector copies, t = x and r = g (lnsrlb, src/lbfgsb.f90:2235-2236), so
 * that the previous iterate survives the trial points written into x and g.  On a bandwidth-bound
 * device those copies are 16 of the 40 bytes per row the one storing pass of an iteration writes.
 * Here the caller hands over TWO buffers for x and TWO for g; nothing is copied: the pair that holds
 * the iterate BECOMES (t, r), the trial point is written to the other pair.
 *   *cur (out) = which pair (0 or 1) this return refers to:
 *     task 'FG...'        evaluate f at x[*cur], write the gradient into g[*cur], pass f as usual;
 *     task 'NEW_X', terminal tasks   x[*cur], g[*cur] are the iterate and its gradient.
 *   task 'START': x0 holds the starting point (the call returns 'FG_START' with *cur = 0).
 * The four buffers must stay the same for the whole run; a run uses either this entry or
 * lbfgsb_hip_setulb_dev, not both (LBFGSB_E_STATE).  A run resumed from lbfgsb_hip_import_state may
 * continue through either entry (here: iterate and gradient in x0 / g0, *cur = 0 until the next step).  Everything else -- task protocol, isave / dsave /
 * lsave, results bit for bit -- as lbfgsb_hip_setulb_dev; lbfgsb_hip_export_state writes the
 * reference's t and r slots from wherever they live.
 * ------------------------------------------------------------------------- */
int lbfgsb_hip_setulb_dev_pp(lbfgsb_hip_ctx *ctx, void *x0, void *x1, const void *l, const void *u,
                             const int32_t *nbd, double *f, void *g0, void *g1, double factr,
                             double pgtol, char *task, int iprint, char *csave, int32_t *lsave,
                             int32_t *isave, double *dsave, int32_t *cur);

/* Stream ordering.  Every kernel of a context runs on ITS stream (the one given to
 * lbfgsb_hip_create, or a private non-blocking stream).  On an 'FG...' re-entry the library
 * reads g (and x, if the caller touched it) on that stream: whatever produced them must be
 * complete, or ordered before the context's stream.  Either evaluate f,g on the context's
 * stream (lbfgsb_hip_get_stream), or call lbfgsb_hip_wait_stream(ctx, producer_stream) before
 * the re-entry -- it records an event on producer_stream and makes the context's stream wait
 * for it, without blocking the host -- or synchronise the producer.  Unless
 * LBFGSB_F_NO_RETURN_SYNC is set, an 'FG...' RETURN has already synchronised the context's
 * stream, so x may be read from any stream. */
void *lbfgsb_hip_get_stream(lbfgsb_hip_ctx *ctx);
int lbfgsb_hip_wait_stream(lbfgsb_hip_ctx *ctx, void *producer_stream);
/* The mirror of lbfgsb_hip_wait_stream, for a caller whose objective runs on ANOTHER stream and who wants no host
 * sync at an 'FG...' return (LBFGSB_F_NO_RETURN_SYNC, LBFGSB_F_DEFER_LNSRCH): an event is recorded on the context's
 * stream behind everything the last call queued -- the storing pass that wrote the trial point included
 * (src/lbfgsb.f90:2255-2272: the point the caller is asked to evaluate) -- and, with make_wait != 0,
 * consumer_stream is made to wait for it (0 is the legacy default stream).  *event_out (may be NULL) receives the
 * hipEvent_t, owned by the context and re-recorded by the next call of this function.  The round trip of such a
 * caller:  setulb -> 'FG...' ; lbfgsb_hip_return_event(ctx, s, 1, NULL) ; f, g evaluated on s ;
 * lbfgsb_hip_f_device / lbfgsb_hip_wait_stream(ctx, s) ; setulb.  Nothing in it blocks the host. */
int lbfgsb_hip_return_event(lbfgsb_hip_ctx *ctx, void *consumer_stream, int make_wait, void **event_out);
/* The caller's objective value as a device scalar (fp64; with several ranks: this rank's part of f, the library
 * adds the parts up): it reaches the host with the next setulb call's first fetch, as the value of a built-in
 * objective evaluated with h_f == NULL does (lbfgsb_hip_objective), and that call stores it in *f -- the 'f'
 * argument of that call is ignored on entry.  order_after != 0: first order the context's stream behind
 * producer_stream (lbfgsb_hip_wait_stream), on which *d_f and g were produced.  Only for an 'FG...' re-entry. */
int lbfgsb_hip_f_device(lbfgsb_hip_ctx *ctx, const void *d_f, void *producer_stream, int order_after);

/* -------------------------------------------------------------------------
 * setulb, host-pointer form: the exact reference signature (what the Fortran
 * shim binds).  x,l,u,g are host arrays of the real kind, nbd/iwa/isave
 * default integers (int32), lsave int32.  The context handle is kept in
 * isave(17:18) (never touched by the reference, src/lbfgsb.f90:250-284),
 * created on task='START' and released when a terminal task is returned (or by
 * lbfgsb_hip_release_host).
 * What crosses PCIe per call, and only then: g host -> device on an 'FG...' entry; x device -> host at START
 * (active's projection), at every 'FG_LNSRCH' return (the trial point) and when the call restored the previous
 * iterate (src/lbfgsb.f90:568-569, 736-737: then g too); `wa`'s t-slot (wa(3n+2mn+11m^2+1 : +n), the previous
 * iterate test/driver3.f90:171-175 reads) when a line search was set up in the call (:2235).  At 'NEW_X' and at
 * the convergence returns the caller already holds x and g: nothing n-long moves.  The caller's x, g and that
 * slot of wa are pinned (hipHostRegister) from START to the end of the run, so the transfers are DMA copies
 * queued on the context's stream; the call returns when they have landed.  Nothing else of wa / iwa is written
 * unless mirror != 0, in which case the full reference layout of wa and iwa is exported on every return.
 * iteration_file may be NULL (-> 'iterate.dat', src/lbfgsb.f90:483-489).
 * ------------------------------------------------------------------------- */
int lbfgsb_hip_setulb_host(int32_t n, int32_t m, void *x, const void *l, const void *u,
                           const int32_t *nbd, void *f, void *g, double factr, double pgtol,
                           void *wa, int32_t *iwa, char *task, int32_t iprint, char *csave,
                           int32_t *lsave, int32_t *isave, void *dsave,
                           const char *iteration_file, int32_t real_bytes, int32_t mirror);
/* The same for a caller whose default INTEGER / LOGICAL kind is int_bytes wide (4 or 8): nbd, iwa,
 * lsave, isave are arrays of that width, n / m / iprint travel as int64.  This is the entry the
 * Fortran module binds (int_bytes = storage_size(1)/8), so that ONE source serves an ordinary build
 * and a -fdefault-integer-8 build -- the build BASELINE.md section 3 calls mandatory for n = 1e8,
 * because the reference's own wa offsets (src/lbfgsb.f90:246-265) overflow a 32-bit integer there.
 * With 8-byte integers isave(1:16) receive those offsets in full; n < 2^31 - 16 on one device.
 * m > LBFGSB_MAX_M is answered the way the reference answers its own argument errors: task =
 * 'ERROR: M > 1024 (LIMIT OF LBFGSB_HIP)', return value 0, no iteration done (the reference itself puts
 * no upper limit on m, :93-97). */
int lbfgsb_hip_setulb_host_ik(int64_t n, int64_t m, void *x, const void *l, const void *u, const void *nbd,
                              void *f, void *g, double factr, double pgtol, void *wa, void *iwa,
                              char *task, int64_t iprint, char *csave, void *lsave, void *isave,
                              void *dsave, const char *iteration_file, int32_t real_bytes,
                              int32_t mirror, int32_t int_bytes);
int lbfgsb_hip_release_host_ik(void *isave, int32_t int_bytes);
/* A caller that leaves its loop without a terminal task FROM the library -- the reference's own
 * driver2/driver3 set task = 'STOP...' and exit (test/driver2.f90:174-195) -- releases the
 * context of the host-pointer form with this call (the Fortran module exports it as
 * lbfgsb_release).  isave(17:18) hold a registry id + tag, never a raw pointer: a stale or
 * garbage isave is refused (LBFGSB_E_STATE), a 'START' over a live id frees the old context
 * first; whatever is still registered at process exit is left to the process teardown (no GPU
 * call is made from a static destructor).  isave(1:16)
 * receive the reference's wa offsets (:250-265), saturated at INT32_MAX instead of wrapped. */
int lbfgsb_hip_release_host(int32_t *isave);
/* Pinning of the caller's arrays by the host-pointer form, for the whole process: mode 1 (default) = START
 * registers x, g and wa's t slot with hipHostRegister (3 n real_bytes: 2.4 GB at n = 1e8) until the run's
 * context goes; mode 0 = never, every transfer is a pageable copy (slower per call, nothing to undo).  Use 0 for
 * short runs of large problems, and for callers that may free x / g / wa of a run they abandoned without
 * lbfgsb_hip_release_host: an array must not be freed while it is registered.  What the library does by itself:
 * a call that arrives with another x, g or wa than START's releases that registration; a START whose arrays
 * overlap the registrations of a run still on the registry drops that (abandoned) run first. */
int lbfgsb_hip_host_pinning(int mode);

/* -------------------------------------------------------------------------
 * Convenience driver around lbfgsb_hip_setulb_dev -- the "high-level wrapper so the user
 * doesn't have to call the reverse communication routine directly" that the reference lists
 * as @todo (src/lbfgsb.f90:36-37).  Runs the loop of test/driver2.f90: evaluates f,g whenever
 * task(1:2)=='FG' -- through `fg` (device pointers in, GLOBAL f out), or, when fg == NULL, the
 * built-in objective `builtin_kind` of lbfgsb_hip_objective -- and stops on a terminal task or
 * when max_iter iterations / max_fg evaluations are reached ('STOP: ...' as the drivers do).
 * On return *f, x, g hold the final point; task/isave/dsave/lsave as after the last setulb.
 * ------------------------------------------------------------------------- */
typedef double (*lbfgsb_fg_fn)(void *user, const void *x_dev, void *g_dev);
int lbfgsb_hip_minimize(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u,
                        const int32_t *nbd, void *g, double factr, double pgtol, int max_iter,
                        int max_fg, int iprint, lbfgsb_fg_fn fg, void *user, int builtin_kind,
                        double *f, char *task, int32_t *lsave, int32_t *isave, double *dsave);

/* -------------------------------------------------------------------------
 * State exchange with the reference's caller-array layout (checkpoint /
 * resume; used by the one-step parity tests).
 * wa: host, length 2mn+5n+11m^2+8m reals; iwa: host int32[3n].
 * export: fills every slot that has a defined meaning at a setulb return
 *   (SURVEY.md appendix B).  import: loads Ws, Wy, z, r, d, t, xp, iwhere,
 *   the free-set membership (from Index(1:nfree)), Sy, Ss, Wt, Wn, Snd, wa8m.
 * With several ranks every rank exports / imports ITS rows (n = n_local in the
 * lengths above; the 2m x 2m matrices are replicated; Index is the local list and
 * Indx2(1) carries this rank's number of free rows): checkpoint at a NEW_X return,
 * resume with the saved task / csave / lsave / isave / dsave on the same partition.
 * Contexts with LBFGSB_F_MIRROR_INDEX are single-rank.
 * ------------------------------------------------------------------------- */
int lbfgsb_hip_export_state(lbfgsb_hip_ctx *ctx, void *wa, int32_t *iwa);
int lbfgsb_hip_import_state(lbfgsb_hip_ctx *ctx, const void *wa, const int32_t *iwa,
                            const int32_t *isave);

/* -------------------------------------------------------------------------
 * Per-kernel entry points (one per row of SURVEY.md 8a), for parity tests
 * and profiling.  All pointers are DEVICE pointers unless named h_*.
 * Reduction results come back in host doubles, already complete across
 * ranks.  Each returns LBFGSB_OK or an error code.
 * ------------------------------------------------------------------------- */

/* projgr, src/lbfgsb.f90:2594-2622 */
int lbfgsb_hip_projgr(lbfgsb_hip_ctx *ctx, const void *x, const void *l, const void *u,
                      const int32_t *nbd, const void *g, double *h_sbgnrm);

/* W'v : out[j] = sum_i Wy(i,col_j) v_i (j < col), out[col+j] = sum_i Ws(i,col_j) v_i,
 * logical column order (head .. head+col-1 mod m).  The WS/WY correction-pair
 * matvec of matupd (:2333-2338), cauchy (:1300-1304) and subsm (:2742-2754).
 * Works on the context's own Ws/Wy. */
int lbfgsb_hip_wtv(lbfgsb_hip_ctx *ctx, const void *v, int col, int head, double *h_out);

/* load host column-major W (n x m each, leading dimension n) into the context */
int lbfgsb_hip_set_w(lbfgsb_hip_ctx *ctx, const void *h_ws, const void *h_wy);

/* formk's inner products from scratch (src/lbfgsb.f90:1756-1851): one masked Gram pass over W
 * with the context's current iwhere (free = iwhere <= 0).  h_out receives 2 col^2 + col sums:
 *   [i(i+1)/2 + j]        i >= j : sum_free Wy_i Wy_j
 *   [T + i(i+1)/2 + j]    i >= j : sum_act  Ws_i Ws_j          (T = col (col+1) / 2)
 *   [2T + i col + j]      all    : sum over (i > j ? active : free) rows of Ws_i Wy_j
 * lbfgsb_hip_set_iwhere loads iwhere (host int32[n_local], the reference's values). */
int lbfgsb_hip_set_iwhere(lbfgsb_hip_ctx *ctx, const int32_t *h_iwhere);
int lbfgsb_hip_formk_gram(lbfgsb_hip_ctx *ctx, int col, int head, double *h_out);

/* -------------------------------------------------------------------------
 * Routine doors: ONE routine of the reference each, on the STATE of the
 * context -- Ws, Wy, Sy, Ss, Wt, WN, WN1 (snd), z, r, d, t, xp, the 8m work vectors, iwhere,
 * Index, Indx2: what lbfgsb_hip_import_state loads and lbfgsb_hip_export_state reads back, in the
 * reference's wa / iwa layout.  A parity test loads the inputs of a routine with import_state,
 * calls the door, and compares export_state (and the door's scalar results) with the CPU twin
 * run on the same arrays (tests/test_gpu_routines.py).  x, l, u, nbd, g and every *_out / *_in
 * vector are DEVICE pointers of n values; everything else is host memory.  Single-rank
 * contexts; every door ends with the stream synchronised.  The doors drive the library's own
 * code (cauchy is the function the iteration calls; freev and matupd share their halves with
 * the iteration; cmprlb, subsm and matupd's n-length sums run through the unfused tile functions
 * that carry m > 32, valid for every m); the fused passes of the hot path are covered call by
 * call by the one-step parity tests.
 * ------------------------------------------------------------------------- */

/* The reference's n-length level-1 BLAS call sites (src/lbfgsb_blas_module.F90:37-277: dcopy / dscal /
 * daxpy / ddot as called from mainlb, lnsrlb, matupd) are terms of the fused passes inside an iteration;
 * the three primitives SURVEY.md 8(b)(4) names have doors of their own all the same.  Device vectors of
 * n_local values of the context's real kind, on the context's stream; no state of the context is read
 * or changed, any number of ranks.
 *   vec_sub    out = a - b            (src/lbfgsb.f90:720-722 d = z - x, :812-816 y = g - r); out may be a or b
 *   vec_scale  v = alpha v            (dscal, :822 s = stp d); alpha is rounded to the context's kind first
 *   dot        *h_result = a'b        (ddot, :816 / :2196 / :2244 / :2335) accumulated in fp64 in the library's
 *              fixed order -- not the reference's groups of five (src/lbfgsb_blas_module.F90:187-202): equal
 *              to it within rounding -- and summed over all ranks of the context (one all-gather, a host
 *              sync); the other two return with the stream synchronised */
int lbfgsb_hip_vec_sub(lbfgsb_hip_ctx *ctx, const void *a, const void *b, void *out);
int lbfgsb_hip_vec_scale(lbfgsb_hip_ctx *ctx, double alpha, void *v);
int lbfgsb_hip_dot(lbfgsb_hip_ctx *ctx, const void *a, const void *b, double *h_result);

/* active, src/lbfgsb.f90:965-1040: x projected onto the box IN PLACE, iwhere initialised;
 * h_flags[0..2] = prjctd, cnstnd, boxed */
int lbfgsb_hip_active(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u, const int32_t *nbd,
                      int32_t *h_flags);
/* errclb, :1601-1643: task (60 chars) is left alone if the input is valid, else 'ERROR: ...',
 * *h_info = -6 / -7 and *h_k = the 1-based index the reference would report */
int lbfgsb_hip_errclb(lbfgsb_hip_ctx *ctx, const void *l, const void *u, const int32_t *nbd, double factr,
                      char *task, int32_t *h_info, int64_t *h_k);
/* cauchy, :1157-1532: reads W, Sy, Wt, iwhere; leaves the Cauchy point in z (and in xcp_out if not
 * NULL), iwhere, p / c / wbp / v in the work vectors (wa8m(1:8m) of export_state), *h_nseg, *h_info */
int lbfgsb_hip_cauchy(lbfgsb_hip_ctx *ctx, const void *x, const void *l, const void *u, const int32_t *nbd,
                      const void *g, double theta, int col, int head, double sbgnrm, void *xcp_out,
                      int32_t *h_nseg, int32_t *h_info);
/* freev, :1980-2059: from iwhere and the previous free set (Index(1:nfree) of the imported state);
 * Index / Indx2 are rebuilt in contexts created with LBFGSB_F_MIRROR_INDEX */
int lbfgsb_hip_freev(lbfgsb_hip_ctx *ctx, int iter, int cnstnd, int updatd, int64_t *h_nfree,
                     int64_t *h_nenter, int64_t *h_ileave, int32_t *h_wrk);
/* formk, :1681-1908: WN1's inner products FROM SCRATCH over the free / active rows of iwhere (the
 * reference keeps them incrementally: the same sums in another order), WN assembled and factorised
 * on the host; *h_info = 0, -1 or -2 */
int lbfgsb_hip_formk(lbfgsb_hip_ctx *ctx, int col, int head, double theta, int32_t *h_info);
/* cmprlb, :1548-1586: r = -Z'(B(xcp - x) + g) from z (= xcp), W, Sy, Wt, c (work vector 2), iwhere;
 * r_out: r scattered to its rows, 0 on the rows that are not free (the reference packs r by Index) */
int lbfgsb_hip_cmprlb(lbfgsb_hip_ctx *ctx, const void *x, const void *g, double theta, int col, int head,
                      int cnstnd, void *r_out, int32_t *h_info);
/* subsm, :2676-2885: r_in as lbfgsb_hip_cmprlb writes it, WN of the state, z = xcp in;
 * z = the subspace minimiser out (and xhat_out if not NULL), xp = xcp, *h_iword, *h_info */
int lbfgsb_hip_subsm(lbfgsb_hip_ctx *ctx, const void *x, const void *l, const void *u, const int32_t *nbd,
                     const void *g, const void *r_in, double theta, int col, int head, void *xhat_out,
                     int32_t *h_iword, int32_t *h_info);
/* lnsrlb, :2174-2275, with mainlb's d = z - x (:720-722) on the first call of an iteration's search.
 * h_sc[8] = fold, gd, gdold, stp, dnorm, dtd, xstep, stpmx; h_ic[7] = iter, ifun, iback, nfgv, info,
 * boxed, cnstnd; task, csave (60 chars), h_isave2[2], h_dsave13[13] as the reference's.  z, d, t, r are
 * the state's; x receives the trial point. */
int lbfgsb_hip_lnsrlb(lbfgsb_hip_ctx *ctx, void *x, const void *l, const void *u, const int32_t *nbd,
                      const void *g, double f, double *h_sc, int32_t *h_ic, char *task, char *csave,
                      int32_t *h_isave2, double *h_dsave13);
/* mainlb :812-824 + matupd, :2291-2346: y = g - r, s = stp d stored in W; Sy, Ss updated.
 * h_ip[4] = iupdat (already counted up, :836), col, head, itail in / out; *h_theta = y'y / dr */
int lbfgsb_hip_matupd(lbfgsb_hip_ctx *ctx, const void *g, double stp, double dr, double dtd, int32_t *h_ip,
                      double *h_theta);

int lbfgsb_hip_sync(lbfgsb_hip_ctx *ctx);

/* built-in device objectives: evaluate f, g on the
 * context's stream.  kind 0 = separable bounded quadratic,
 * kind 1 = extended Rosenbrock (test/driver1.f90:274-289; sharded: the 1-element halo x(row0-1), x(row0+n) is all-gathered through the communicator).
 * *h_f receives the GLOBAL value (reduced over ranks).  h_f == NULL defers it: the value stays
 * on the device and the NEXT lbfgsb_hip_setulb_dev call on this context -- which must be the
 * 'FG...' re-entry for this evaluation -- brings it over with the sums of its own first pass
 * (one host sync and one all-reduce less per evaluation) and stores it through its f argument. */
int lbfgsb_hip_objective(lbfgsb_hip_ctx *ctx, int kind, const void *x, void *g, double *h_f);

/* Per-context options for measurements and tests (A/B timings of fallback paths, forcing rarely
 * taken routes).  Nothing in the library reads tuning switches from the environment: a context
 * behaves as created unless this is called.  Names (value):
 *   "two_pass" (0/1)        the two-pass iteration with W'Z r in closed form; 0 = always the
 *                           cmprlb_wtv pass (three passes over W)
 *   "two_pass_maxcol" (0..32)  largest col that takes the two-pass iteration (default 32; beyond 21 pairs the update
 *                           pass runs as two launches over half of the columns each; 20 = three passes there)
 *   "lean" (0/1)            z and d = x - t left implicit by the storing pass
 *   "spec_capture" (0/1)    the update pass hands the next walk's first breakpoints over
 *   "pg_min" (count)        LBFGSB_F_PARALLEL_GCP: walks with more breakpoints in reach than this
 *                           go to the sort + scans
 *   "exact_always" (0/1)    every Cauchy walk in the reference's heap order from its start (tests)
 *   "defer_lnsrch" (0/1)    LBFGSB_F_DEFER_LNSRCH switched on / off for the iterations that follow (the
 *                           caller must honour that flag's contract)
 *   "spin" (0/1)            results of a phase reach the host through mapped memory + a polled sequence
 *                           word (default) / through a D2H copy + hipStreamSynchronize
 *   "fold_finalize" (0/1)   reductions nobody waits for yet (a deferred f, a deferred set-up) leave their
 *                           finalize to the next launch (default 1)
 *   "eager_patch" (0/1)     formk's patch sums are queued behind freev's counting pass and fetched with its
 *                           counts (default 1) / after a host round trip of their own
 *   "spec_freev" (0/1)      while the free set changes from iteration to iteration, freev's counting pass and
 *                           the patch are queued speculatively behind the evaluation of a trial point and
 *                           used if the point is accepted and the next walk fixes no row (default 0: measured,
 *                           does not pay)
 *   "skip_reuse" (0/1)      after a skipped BFGS update, cauchy's n-loop sums come from the pass that evaluated
 *                           the accepted point (+ a one-column scan when the memory is full) (default 1) /
 *                           from a scan over all of W
 *   "wide_fused" (0/1)      m > 32: matupd's, cauchy's and formk's sums from the update pass (split over the columns,
 *                           one pass over W) in front of the unfused subspace steps (default 1) / all unfused
 *   "wide_closed" (0/1)     m > 32: W'Z r in closed form and cmprlb's + subsm's updates of r as one pass over W
 *                           (default 1) / a W'r pass and two updates
 *   "wide_tail" (0/1)       m > 32: cmprlb's start and subsm's projected step + the line-search set-up folded into the first /
 *                           last tile of that pass (default 1) / as kernels of their own
 *   "win_slack" (>= 0)      the first window of a breakpoint walk asks (1 + win_slack) x as far ahead as the walk needs
 *                           when it starts (default 0.25; 0: exactly as far -- a second window pass usually follows)
 *   "spec_trial2" (0/1)     the SECOND trial point of a line search (an interpolated step after a rejected first one) is
 *                           evaluated by the update pass too, whose sums serve the NEW_X entry if it is accepted
 *                           (default 1) / by the two-sum evaluation kernel, the update pass follows at NEW_X
 *   "wide_one" (0/1)        m > 32, col <= 96: that pass as ONE launch over all columns, the pending pair committed by
 *                           it (default 1) / one launch per tile of 32 columns behind pair_commit
 *   "wide_incr" (0/1)       m > 32: formk adds the new pair's row and column to WN1 while no row changes status
 *                           (default 1) / from scratch whenever it runs
 *   "nt" (0/1)              nontemporal loads in the passes over W (default: by the size of W)
 *   "uniform_bounds" (0/1)  detect bound arrays that hold one value each (lbfgsb_hip_uniform_bounds)
 *   "dict_bounds" (0/1)     dictionary-code bound arrays with <= 8 distinct values each (same place; default 1)
 *   "bounds_check" (0..)    compare the caller's l, u, nbd with the context's snapshot after the first iteration and
 *                           then every k-th (default 32; 0 = never): 'ERROR: BOUNDS CHANGED DURING RUN'
 *   "wgrid" (0..2047)       workgroups of the passes over W (default 0: what is resident for the kernel
 *                           launched, 256 ... 768)
 *   "pipe" (-1/0/1)         two trips of loads in flight per wave: default rule (m = 20, fp32 m = 10) / off /
 *                           the default rule again (the other shapes are not compiled with it)
 *   "pair" (0/1/2)          MC = 20 update pass: lane pairs share accumulators (off / 1 trip / 2 trips)
 *   "split" (10/20)         update pass with formk's new-row sums: split over the columns beyond 20 old pairs into
 *                           parts of <= 16 (default) / beyond 10 into parts of <= 10 (measurement: slower at m = 20)
 *   "gram_rows" (0/1)       formk from scratch with the LDS-slab kernel instead of the quad kernel
 * Returns LBFGSB_E_ARG for an unknown name or a value out of range. */
int lbfgsb_hip_set_option(lbfgsb_hip_ctx *ctx, const char *name, double value);

/* Uniform and few-valued bounds.  Bound arrays that hold ONE value each -- the box [a, b]^n, x >= 0 -- are the
 * common case, and on a bandwidth-bound device streaming 2 x 8 + 1 constant bytes per row through each of the
 * two passes over W of an iteration is 8 % of its traffic.  At task 'START' the pass that validates the bounds
 * (errclb, src/lbfgsb.f90:1601-1643) also checks, bit for bit, whether every l_i equals l_1, every u_i equals
 * u_1, every nbd_i equals nbd_1 (among this rank's rows); the passes over W then read the value from a 64-byte
 * buffer instead of the array -- same arithmetic, same results bit for bit.
 * Arrays with a FEW distinct values -- the reference's own test box, l alternating 1 / -100 with u = 100
 * (test/driver3.f90:102-120); a box with some variables on another box -- are dictionary-coded: if l and u hold
 * at most 8 distinct values each (bit patterns; found by a few probing passes at START, identical on all ranks),
 * the one-byte copy of nbd the passes stream anyway carries  nbd | l-index << 2 | u-index << 5  and the values
 * come from two 8-entry tables held in LDS: 1 byte per row instead of 17.  Same values, same arithmetic.
 * *mask: bit 0 = l, bit 1 = u, bit 2 = nbd is treated as uniform in the current run; bit 3 (with bits 0 and 1):
 * l and u are dictionary-coded.  Passing OTHER array pointers than at 'START' switches the corresponding bits
 * off (any of the three for the dictionary).  Options "uniform_bounds" = 0 / "dict_bounds" = 0
 * (lbfgsb_hip_set_option, before START) disable the detection / the dictionary. */
int lbfgsb_hip_uniform_bounds(lbfgsb_hip_ctx *ctx, int32_t *mask);

/* Profiling clocks, counters and bare-kernel timing doors (bench.py, profiles/scripts, tests) are declared in
 * lbfgsb_hip_debug.h: measurement instruments of the same library, not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif
