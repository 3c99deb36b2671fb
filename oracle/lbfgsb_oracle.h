/*
 * lbfgsb_oracle.h -- CPU ORACLE (test infrastructure only, NOT the product).
 *
 * Plain-C restatement of the reference's L-BFGS-B reverse-communication path
 * (jacobwilliams/lbfgsb, src/lbfgsb.f90 + lbfgsb_blas_module.F90 +
 * lbfgsb_linpack_module.f90).  Every function cites the reference file:line it
 * follows.  Summation order, branch order and operation order are kept
 * identical to the Fortran so that a build with `-O2 -ffp-contract=off` is
 * bit-for-bit comparable with the reference built by amdflang -O2.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or load this.  The product (lbfgsb_amd/) never does.
 *
 * Parity pinning: validated against the real reference (oracle/_ref, built
 * from /root/reference by oracle/Makefile) and against the committed
 * fixtures in tests/golden/ (see tests/test_oracle_golden.py).
 *
 * Array conventions: all index arrays hold 1-based variable numbers exactly
 * as the Fortran does, so that snapshots of `iwa` compare directly.
 * `wa` layout = reference setulb (src/lbfgsb.f90:250-265) with 64-bit
 * offsets (the reference overflows int32 at n=1e8).
 */
#ifndef LBFGSB_ORACLE_H
#define LBFGSB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef LBO_REAL32
typedef float lbo_real;
#else
typedef double lbo_real;
#endif

/* src/lbfgsb.f90:88-286  setulb.  task/csave are 60-char blank padded,
 * lsave are C ints (0/1).  isave[44], dsave[29] as documented at :194-242. */
void lbo_setulb(int n, int m, lbo_real *x, const lbo_real *l, const lbo_real *u,
                const int *nbd, lbo_real *f, lbo_real *g, lbo_real factr,
                lbo_real pgtol, lbo_real *wa, int *iwa, char *task, int iprint,
                char *csave, int *lsave, int *isave, lbo_real *dsave);

/* wa length and slot offsets (0-based) for given n, m: :250-265 */
int64_t lbo_wa_len(int64_t n, int64_t m);
void lbo_wa_offsets(int64_t n, int64_t m, int64_t off[13]);
/* off[] order: ws wy sy ss wt wn snd z r d t xp wa8m */

/* ---- phase routines, exposed for white-box kernel parity tests ---- */

/* :965-1040 */
void lbo_active(int n, const lbo_real *l, const lbo_real *u, const int *nbd,
                lbo_real *x, int *iwhere, int *prjctd, int *cnstnd, int *boxed,
                int *nbdd);
/* :1601-1643; returns info, writes task on error, *k = offending index */
void lbo_errclb(int n, int m, lbo_real factr, const lbo_real *l,
                const lbo_real *u, const int *nbd, char *task, int *info, int *k);
/* :2594-2622 */
lbo_real lbo_projgr(int n, const lbo_real *l, const lbo_real *u, const int *nbd,
                    const lbo_real *x, const lbo_real *g);
/* :1057-1123 */
void lbo_bmv(int m, const lbo_real *sy, const lbo_real *wt, int col,
             const lbo_real *v, lbo_real *p, int *info);
/* :1157-1532 */
void lbo_cauchy(int n, const lbo_real *x, const lbo_real *l, const lbo_real *u,
                const int *nbd, const lbo_real *g, int *iorder, int *iwhere,
                lbo_real *t, lbo_real *d, lbo_real *xcp, int m,
                const lbo_real *wy, const lbo_real *ws, const lbo_real *sy,
                const lbo_real *wt, lbo_real theta, int col, int head,
                lbo_real *p, lbo_real *c, lbo_real *wbp, lbo_real *v, int *nseg,
                lbo_real sbgnrm, int *info, lbo_real epsmch);
/* :2079-2157 */
void lbo_hpsolb(int n, lbo_real *t, int *iorder, int iheap);
/* :1980-2059 */
void lbo_freev(int n, int *nfree, int *index, int *nenter, int *ileave,
               int *indx2, const int *iwhere, int *wrk, int updatd, int cnstnd,
               int iter);
/* :1681-1908 */
void lbo_formk(int n, int nsub, const int *ind, int nenter, int ileave,
               const int *indx2, int iupdat, int updatd, lbo_real *wn,
               lbo_real *wn1, int m, const lbo_real *ws, const lbo_real *wy,
               const lbo_real *sy, lbo_real theta, int col, int head, int *info);
/* :1548-1586 */
void lbo_cmprlb(int n, int m, const lbo_real *x, const lbo_real *g,
                const lbo_real *ws, const lbo_real *wy, const lbo_real *sy,
                const lbo_real *wt, const lbo_real *z, lbo_real *r, lbo_real *wa,
                const int *index, lbo_real theta, int col, int head, int nfree,
                int cnstnd, int *info);
/* :2676-2885 */
void lbo_subsm(int n, int m, int nsub, const int *ind, const lbo_real *l,
               const lbo_real *u, const int *nbd, lbo_real *x, lbo_real *d,
               lbo_real *xp, const lbo_real *ws, const lbo_real *wy,
               lbo_real theta, const lbo_real *xx, const lbo_real *gg, int col,
               int head, int *iword, lbo_real *wv, const lbo_real *wn, int *info);
/* :2174-2275 */
void lbo_lnsrlb(int n, const lbo_real *l, const lbo_real *u, const int *nbd,
                lbo_real *x, lbo_real f, lbo_real *fold, lbo_real *gd,
                lbo_real *gdold, const lbo_real *g, const lbo_real *d,
                lbo_real *r, lbo_real *t, const lbo_real *z, lbo_real *stp,
                lbo_real *dnorm, lbo_real *dtd, lbo_real *xstep,
                lbo_real *stpmx, int iter, int *ifun, int *iback, int *nfgv,
                int *info, char *task, int boxed, int cnstnd, char *csave,
                int *isave2, lbo_real *dsave13);
/* :2291-2346 */
void lbo_matupd(int n, int m, lbo_real *ws, lbo_real *wy, lbo_real *sy,
                lbo_real *ss, const lbo_real *d, const lbo_real *r, int *itail,
                int iupdat, int *col, int *head, lbo_real *theta, lbo_real rr,
                lbo_real dr, lbo_real stp, lbo_real dtd);
/* :1926-1963 */
void lbo_formt(int m, lbo_real *wt, const lbo_real *sy, const lbo_real *ss,
               int col, lbo_real theta, int *info);
/* :2942-3198 */
void lbo_dcsrch(lbo_real *f, lbo_real *g, lbo_real *stp, lbo_real ftol,
                lbo_real gtol, lbo_real xtol, lbo_real stpmin, lbo_real stpmax,
                char *task, int *isave, lbo_real *dsave);
/* lbfgsb_linpack_module.f90:30-67, 87-165 */
void lbo_dpofa(lbo_real *a, int lda, int n, int *info);
void lbo_dtrsl(const lbo_real *t, int ldt, int n, lbo_real *b, int job, int *info);
/* lbfgsb_blas_module.F90:165-222 */
lbo_real lbo_ddot(int64_t n, const lbo_real *dx, const lbo_real *dy);

/* test instrumentation: [0] = times the subsm backtracking branch (:2830-2879) ran,
 * [1] = memory refreshes in mainlb.  Lets tests locate calls that take rare branches. */
extern long lbo_branch_count[4];

/* Synthetic objectives used by bench/tests (these are
 * definitions from SURVEY.md section 8d, not reference code).  i0 = 0-based
 * global index of element 0 (for sharded evaluation).  Returns f. */
lbo_real lbo_quadratic_fg(int64_t n, int64_t i0, const lbo_real *x, lbo_real *g);
/* test/driver1.f90:274-289 (extended Rosenbrock) */
lbo_real lbo_rosenbrock_fg(int64_t n, const lbo_real *x, lbo_real *g);

#ifdef __cplusplus
}
#endif
#endif
