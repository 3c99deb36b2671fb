/*
 * lbfgsb_oracle.c -- CPU ORACLE (test infrastructure only, NOT the product).
 *
 * Plain-C restatement of the algorithm in the reference
 * /root/reference/src/lbfgsb.f90 (setulb/mainlb and everything they call),
 * lbfgsb_blas_module.F90 and lbfgsb_linpack_module.f90.  See
 * lbfgsb_oracle.h for the rules.  Text output (prn1lb/prn2lb/prn3lb,
 * src/lbfgsb.f90:2363-2579) is not restated: the oracle is silent for every
 * iprint; the iteration-file unit slot isave(24) is left 0.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * Parity pinned by tests/test_oracle_golden.py: fixtures produced by the real
 * reference, and live runs against oracle/_ref when it is built.
 */
#include "lbfgsb_oracle.h"

#include <math.h>
#include <string.h>
#include <time.h>

typedef lbo_real real;

/* test instrumentation: how often rare branches were taken (read by tests to pick cases that
 * exercise them); [0] = subsm backtracking branch (:2830-2879), [1] = memory refreshes in mainlb */
long lbo_branch_count[4] = {0, 0, 0, 0};

#ifdef LBO_REAL32
#define RSQRT(x) sqrtf(x)
#define RABS(x) fabsf(x)
#define REPS 1.1920929e-07f
#else
#define RSQRT(x) sqrt(x)
#define RABS(x) fabs(x)
#define REPS 2.220446049250313e-16
#endif

#define ZERO ((real)0)
#define ONE ((real)1)
#define TWO ((real)2)
#define THREE ((real)3)

static real rmax(real a, real b) { return a > b ? a : b; } /* Fortran max */
static real rmin(real a, real b) { return a < b ? a : b; }

/* ---------- 60-char blank padded task strings ---------- */
static void task_set(char *t, const char *s) {
  size_t k = strlen(s);
  if (k > 60) k = 60;
  memcpy(t, s, k);
  memset(t + k, ' ', 60 - k);
}
static int task_pre(const char *t, const char *s) { return strncmp(t, s, strlen(s)) == 0; }
static int task_eq(const char *t, const char *s) {
  size_t k = strlen(s), i;
  if (strncmp(t, s, k) != 0) return 0;
  for (i = k; i < 60; i++)
    if (t[i] != ' ') return 0;
  return 1;
}

static double cpu_now(void) { return (double)clock() / (double)CLOCKS_PER_SEC; }

/* ---------- level-1 BLAS: lbfgsb_blas_module.F90 ---------- */

/* :165-222 (unit stride path; clean-up loop FIRST, then groups of 5 added
 * left to right -- i.e. a plain sequential sum) */
real lbo_ddot(int64_t n, const real *dx, const real *dy) {
  real dtemp = ZERO;
  int64_t i, m;
  if (n <= 0) return ZERO;
  m = n % 5;
  for (i = 0; i < m; i++) dtemp = dtemp + dx[i] * dy[i];
  if (n < 5) return dtemp;
  for (i = m; i < n; i += 5)
    dtemp = dtemp + dx[i] * dy[i] + dx[i + 1] * dy[i + 1] + dx[i + 2] * dy[i + 2] +
            dx[i + 3] * dy[i + 3] + dx[i + 4] * dy[i + 4];
  return dtemp;
}
/* :37-89 */
static void daxpy(int64_t n, real da, const real *dx, real *dy) {
  int64_t i;
  if (n <= 0) return;
  if (da == ZERO) return;
  for (i = 0; i < n; i++) dy[i] = dy[i] + da * dx[i];
}
/* :100-154 */
static void dcopy(int64_t n, const real *dx, real *dy) {
  if (n > 0) memmove(dy, dx, (size_t)n * sizeof(real));
}
/* :233-277 */
static void dscal(int64_t n, real da, real *dx) {
  int64_t i;
  for (i = 0; i < n; i++) dx[i] = da * dx[i];
}

/* ---------- LINPACK: lbfgsb_linpack_module.f90 ---------- */
#define A_(i, j) a[((i)-1) + (int64_t)((j)-1) * lda]

/* :30-67 */
void lbo_dpofa(real *a, int lda, int n, int *info) {
  int j, k;
  real s, t;
  for (j = 1; j <= n; j++) {
    *info = j;
    s = ZERO;
    for (k = 1; k <= j - 1; k++) {
      t = A_(k, j) - lbo_ddot(k - 1, &A_(1, k), &A_(1, j));
      t = t / A_(k, k);
      A_(k, j) = t;
      s = s + t * t;
    }
    s = A_(j, j) - s;
    if (s <= ZERO) return;
    A_(j, j) = RSQRT(s);
  }
  *info = 0;
}
#undef A_

#define T_(i, j) t[((i)-1) + (int64_t)((j)-1) * ldt]
/* :87-165; b is 1-based in the comments, 0-based here */
void lbo_dtrsl(const real *t, int ldt, int n, real *b, int job, int *info) {
  int j, jj, kase;
  real temp;
  for (*info = 1; *info <= n; (*info)++)
    if (T_(*info, *info) == ZERO) return;
  *info = 0;
  kase = 1;
  if (job % 10 != 0) kase = 2;
  if ((job % 100) / 10 != 0) kase += 2;
  switch (kase) {
    case 1: /* t*x=b, t lower */
      b[0] = b[0] / T_(1, 1);
      for (j = 2; j <= n; j++) {
        temp = -b[j - 2];
        daxpy(n - j + 1, temp, &T_(j, j - 1), &b[j - 1]);
        b[j - 1] = b[j - 1] / T_(j, j);
      }
      break;
    case 2: /* t*x=b, t upper */
      b[n - 1] = b[n - 1] / T_(n, n);
      for (jj = 2; jj <= n; jj++) {
        j = n - jj + 1;
        temp = -b[j];
        daxpy(j, temp, &T_(1, j + 1), &b[0]);
        b[j - 1] = b[j - 1] / T_(j, j);
      }
      break;
    case 3: /* trans(t)*x=b, t lower */
      b[n - 1] = b[n - 1] / T_(n, n);
      for (jj = 2; jj <= n; jj++) {
        j = n - jj + 1;
        b[j - 1] = b[j - 1] - lbo_ddot(jj - 1, &T_(j + 1, j), &b[j]);
        b[j - 1] = b[j - 1] / T_(j, j);
      }
      break;
    case 4: /* trans(t)*x=b, t upper */
      b[0] = b[0] / T_(1, 1);
      for (j = 2; j <= n; j++) {
        b[j - 1] = b[j - 1] - lbo_ddot(j - 1, &T_(1, j), &b[0]);
        b[j - 1] = b[j - 1] / T_(j, j);
      }
      break;
  }
}
#undef T_

/* ---------- wa partition: src/lbfgsb.f90:250-265 ---------- */
int64_t lbo_wa_len(int64_t n, int64_t m) { return 2 * m * n + 5 * n + 11 * m * m + 8 * m; }
void lbo_wa_offsets(int64_t n, int64_t m, int64_t off[13]) {
  off[0] = 0;                    /* ws   m*n  */
  off[1] = off[0] + m * n;       /* wy   m*n  */
  off[2] = off[1] + m * n;       /* sy   m^2  */
  off[3] = off[2] + m * m;       /* ss   m^2  */
  off[4] = off[3] + m * m;       /* wt   m^2  */
  off[5] = off[4] + m * m;       /* wn   4m^2 */
  off[6] = off[5] + 4 * m * m;   /* snd  4m^2 */
  off[7] = off[6] + 4 * m * m;   /* z    n    */
  off[8] = off[7] + n;           /* r    n    */
  off[9] = off[8] + n;           /* d    n    */
  off[10] = off[9] + n;          /* t    n    */
  off[11] = off[10] + n;         /* xp   n    */
  off[12] = off[11] + n;         /* wa   8m   */
}

/* ---------- src/lbfgsb.f90:965-1040 active ---------- */
void lbo_active(int n, const real *l, const real *u, const int *nbd, real *x, int *iwhere,
                int *prjctd, int *cnstnd, int *boxed, int *nbdd_out) {
  int i, nbdd = 0;
  *prjctd = 0;
  *cnstnd = 0;
  *boxed = 1;
  for (i = 0; i < n; i++) {
    if (nbd[i] > 0) {
      if (nbd[i] <= 2 && x[i] <= l[i]) {
        if (x[i] < l[i]) {
          *prjctd = 1;
          x[i] = l[i];
        }
        nbdd++;
      } else if (nbd[i] >= 2 && x[i] >= u[i]) {
        if (x[i] > u[i]) {
          *prjctd = 1;
          x[i] = u[i];
        }
        nbdd++;
      }
    }
  }
  for (i = 0; i < n; i++) {
    if (nbd[i] != 2) *boxed = 0;
    if (nbd[i] == 0) {
      iwhere[i] = -1;
    } else {
      *cnstnd = 1;
      if (nbd[i] == 2 && u[i] - l[i] <= ZERO)
        iwhere[i] = 3;
      else
        iwhere[i] = 0;
    }
  }
  if (nbdd_out) *nbdd_out = nbdd;
}

/* ---------- :1601-1643 errclb ---------- */
void lbo_errclb(int n, int m, real factr, const real *l, const real *u, const int *nbd,
                char *task, int *info, int *k) {
  int i;
  if (n <= 0) task_set(task, "ERROR: N <= 0");
  if (m <= 0) task_set(task, "ERROR: M <= 0");
  if (factr < ZERO) task_set(task, "ERROR: FACTR < 0");
  *k = 0;
  for (i = 1; i <= n; i++) {
    if (nbd[i - 1] < 0 || nbd[i - 1] > 3) {
      task_set(task, "ERROR: INVALID NBD");
      *info = -6;
      *k = i;
    }
    if (nbd[i - 1] == 2) {
      if (l[i - 1] > u[i - 1]) {
        task_set(task, "ERROR: NO FEASIBLE SOLUTION");
        *info = -7;
        *k = i;
      }
    }
  }
}

/* ---------- :2594-2622 projgr ---------- */
real lbo_projgr(int n, const real *l, const real *u, const int *nbd, const real *x,
                const real *g) {
  int i;
  real gi, sbgnrm = ZERO;
  for (i = 0; i < n; i++) {
    gi = g[i];
    if (nbd[i] != 0) {
      if (gi < ZERO) {
        if (nbd[i] >= 2) gi = rmax(x[i] - u[i], gi);
      } else {
        if (nbd[i] <= 2) gi = rmin(x[i] - l[i], gi);
      }
    }
    sbgnrm = rmax(sbgnrm, RABS(gi));
  }
  return sbgnrm;
}

/* ---------- :1057-1123 bmv ---------- */
#define SY(i, j) sy[((i)-1) + (int64_t)((j)-1) * m]
#define SS(i, j) ss[((i)-1) + (int64_t)((j)-1) * m]
#define WT(i, j) wt[((i)-1) + (int64_t)((j)-1) * m]
void lbo_bmv(int m, const real *sy, const real *wt, int col, const real *v, real *p,
             int *info) {
  int i, k, i2;
  real sum;
  *info = 0;
  if (col == 0) return;
  p[col] = v[col];
  for (i = 2; i <= col; i++) {
    i2 = col + i;
    sum = ZERO;
    for (k = 1; k <= i - 1; k++) sum = sum + SY(i, k) * v[k - 1] / SY(k, k);
    p[i2 - 1] = v[i2 - 1] + sum;
  }
  lbo_dtrsl(wt, m, col, &p[col], 11, info);
  if (*info != 0) return;
  for (i = 1; i <= col; i++) p[i - 1] = v[i - 1] / RSQRT(SY(i, i));
  lbo_dtrsl(wt, m, col, &p[col], 1, info);
  if (*info != 0) return;
  for (i = 1; i <= col; i++) p[i - 1] = -p[i - 1] / RSQRT(SY(i, i));
  for (i = 1; i <= col; i++) {
    sum = ZERO;
    for (k = i + 1; k <= col; k++) sum = sum + SY(k, i) * p[col + k - 1] / SY(i, i);
    p[i - 1] = p[i - 1] + sum;
  }
}

/* ---------- :2079-2157 hpsolb (t, iorder 1-based in comments) ---------- */
void lbo_hpsolb(int n, real *t, int *iorder, int iheap) {
  int i, j, k, indxin, indxou;
  real ddum, out;
#define T1(i) t[(i)-1]
#define IO1(i) iorder[(i)-1]
  if (iheap == 0) {
    for (k = 2; k <= n; k++) {
      ddum = T1(k);
      indxin = IO1(k);
      i = k;
      for (;;) {
        if (i > 1) {
          j = i / 2;
          if (ddum < T1(j)) {
            T1(i) = T1(j);
            IO1(i) = IO1(j);
            i = j;
            continue;
          }
        }
        break;
      }
      T1(i) = ddum;
      IO1(i) = indxin;
    }
  }
  if (n > 1) {
    i = 1;
    out = T1(1);
    indxou = IO1(1);
    ddum = T1(n);
    indxin = IO1(n);
    for (;;) {
      j = i + i;
      if (j <= n - 1) {
        if (T1(j + 1) < T1(j)) j = j + 1;
        if (T1(j) < ddum) {
          T1(i) = T1(j);
          IO1(i) = IO1(j);
          i = j;
          continue;
        }
      }
      break;
    }
    T1(i) = ddum;
    IO1(i) = indxin;
    T1(n) = out;
    IO1(n) = indxou;
  }
#undef T1
#undef IO1
}

/* ---------- :1157-1532 cauchy ---------- */
#define WS(i, j) ws[((int64_t)(i)-1) + (int64_t)((j)-1) * n]
#define WY(i, j) wy[((int64_t)(i)-1) + (int64_t)((j)-1) * n]
void lbo_cauchy(int n, const real *x, const real *l, const real *u, const int *nbd,
                const real *g, int *iorder, int *iwhere, real *t, real *d, real *xcp, int m,
                const real *wy, const real *ws, const real *sy, const real *wt, real theta,
                int col, int head, real *p, real *c, real *wbp, real *v, int *nseg,
                real sbgnrm, int *info, real epsmch) {
  int xlower, xupper, bnded;
  int i, j, col2, nfree, nbreak, pointr, ibp = 0, nleft, ibkmin, iter;
  real f1, f2, dt, dtm, tsum, dibp, zibp, dibp2, bkmin, tu = ZERO, tl = ZERO, wmc, wmp, wmw,
                                                         tj, tj0, neggi, f2_org;

  if (sbgnrm <= ZERO) { /* :1245-1249 */
    dcopy(n, x, xcp);
    return;
  }
  bnded = 1;
  nfree = n + 1;
  nbreak = 0;
  ibkmin = 0;
  bkmin = ZERO;
  col2 = 2 * col;
  f1 = ZERO;
  for (i = 0; i < col2; i++) p[i] = ZERO;

  /* :1270-1330 */
  for (i = 1; i <= n; i++) {
    neggi = -g[i - 1];
    if (iwhere[i - 1] != 3 && iwhere[i - 1] != -1) {
      if (nbd[i - 1] <= 2) tl = x[i - 1] - l[i - 1];
      if (nbd[i - 1] >= 2) tu = u[i - 1] - x[i - 1];
      xlower = nbd[i - 1] <= 2 && tl <= ZERO;
      xupper = nbd[i - 1] >= 2 && tu <= ZERO;
      iwhere[i - 1] = 0;
      if (xlower) {
        if (neggi <= ZERO) iwhere[i - 1] = 1;
      } else if (xupper) {
        if (neggi >= ZERO) iwhere[i - 1] = 2;
      } else {
        if (RABS(neggi) <= ZERO) iwhere[i - 1] = -3;
      }
    }
    pointr = head;
    if (iwhere[i - 1] != 0 && iwhere[i - 1] != -1) {
      d[i - 1] = ZERO;
    } else {
      d[i - 1] = neggi;
      f1 = f1 - neggi * neggi;
      for (j = 1; j <= col; j++) {
        p[j - 1] = p[j - 1] + WY(i, pointr) * neggi;
        p[col + j - 1] = p[col + j - 1] + WS(i, pointr) * neggi;
        pointr = pointr % m + 1;
      }
      if (nbd[i - 1] <= 2 && nbd[i - 1] != 0 && neggi < ZERO) {
        nbreak = nbreak + 1;
        iorder[nbreak - 1] = i;
        t[nbreak - 1] = tl / (-neggi);
        if (nbreak == 1 || t[nbreak - 1] < bkmin) {
          bkmin = t[nbreak - 1];
          ibkmin = nbreak;
        }
      } else if (nbd[i - 1] >= 2 && neggi > ZERO) {
        nbreak = nbreak + 1;
        iorder[nbreak - 1] = i;
        t[nbreak - 1] = tu / neggi;
        if (nbreak == 1 || t[nbreak - 1] < bkmin) {
          bkmin = t[nbreak - 1];
          ibkmin = nbreak;
        }
      } else {
        nfree = nfree - 1;
        iorder[nfree - 1] = i;
        if (RABS(neggi) > ZERO) bnded = 0;
      }
    }
  }

  if (theta != ONE) dscal(col, theta, &p[col]); /* :1337 */
  dcopy(n, x, xcp);                             /* :1341 */
  if (nbreak == 0 && nfree == n + 1) return;    /* :1343-1347 */
  for (j = 0; j < col2; j++) c[j] = ZERO;

  f2 = -theta * f1; /* :1357-1363 */
  f2_org = f2;
  if (col > 0) {
    lbo_bmv(m, sy, wt, col, p, v, info);
    if (*info != 0) return;
    f2 = f2 - lbo_ddot(col2, v, p);
  }
  dtm = -f1 / f2;
  tsum = ZERO;
  *nseg = 1;

  if (nbreak != 0) {
    nleft = nbreak;
    iter = 1;
    tj = ZERO;
    for (;;) { /* :1378-1497 */
      tj0 = tj;
      if (iter == 1) {
        tj = bkmin;
        ibp = iorder[ibkmin - 1];
      } else {
        if (iter == 2) {
          if (ibkmin != nbreak) {
            t[ibkmin - 1] = t[nbreak - 1];
            iorder[ibkmin - 1] = iorder[nbreak - 1];
          }
        }
        lbo_hpsolb(nleft, t, iorder, iter - 2);
        tj = t[nleft - 1];
        ibp = iorder[nleft - 1];
      }
      dt = tj - tj0;
      if (dtm < dt) break; /* :1416 */

      tsum = tsum + dt;
      nleft = nleft - 1;
      iter = iter + 1;
      dibp = d[ibp - 1];
      d[ibp - 1] = ZERO;
      if (dibp > ZERO) {
        zibp = u[ibp - 1] - x[ibp - 1];
        xcp[ibp - 1] = u[ibp - 1];
        iwhere[ibp - 1] = 2;
      } else {
        zibp = l[ibp - 1] - x[ibp - 1];
        xcp[ibp - 1] = l[ibp - 1];
        iwhere[ibp - 1] = 1;
      }
      if (nleft == 0 && nbreak == n) { /* :1436-1442 */
        dtm = dt;
        if (col > 0) daxpy(col2, dtm, p, c);
        return;
      }
      *nseg = *nseg + 1;
      dibp2 = dibp * dibp;
      f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp; /* :1452-1453 */
      f2 = f2 - theta * dibp2;
      if (col > 0) {
        daxpy(col2, dt, p, c);
        pointr = head;
        for (j = 1; j <= col; j++) {
          wbp[j - 1] = WY(ibp, pointr);
          wbp[col + j - 1] = theta * WS(ibp, pointr);
          pointr = pointr % m + 1;
        }
        lbo_bmv(m, sy, wt, col, wbp, v, info);
        if (*info != 0) return;
        wmc = lbo_ddot(col2, c, v);
        wmp = lbo_ddot(col2, p, v);
        wmw = lbo_ddot(col2, wbp, v);
        daxpy(col2, -dibp, wbp, p);
        f1 = f1 + dibp * wmc;
        f2 = f2 + TWO * dibp * wmp - dibp2 * wmw;
      }
      f2 = rmax(epsmch * f2_org, f2); /* :1483 */
      if (nleft > 0) {
        dtm = -f1 / f2;
      } else if (bnded) {
        f1 = ZERO;
        f2 = ZERO;
        dtm = ZERO;
        break;
      } else {
        dtm = -f1 / f2;
        break;
      }
    }
  }
  if (dtm <= ZERO) dtm = ZERO; /* :1509 */
  tsum = tsum + dtm;
  daxpy(n, tsum, d, xcp);              /* :1515 */
  if (col > 0) daxpy(col2, dtm, p, c); /* :1526 */
}

/* ---------- :1548-1586 cmprlb ---------- */
void lbo_cmprlb(int n, int m, const real *x, const real *g, const real *ws, const real *wy,
                const real *sy, const real *wt, const real *z, real *r, real *wa,
                const int *index, real theta, int col, int head, int nfree, int cnstnd,
                int *info) {
  int i, j, k, pointr;
  real a1, a2;
  if (!cnstnd && col > 0) {
    for (i = 0; i < n; i++) r[i] = -g[i];
  } else {
    for (i = 1; i <= nfree; i++) {
      k = index[i - 1];
      r[i - 1] = -theta * (z[k - 1] - x[k - 1]) - g[k - 1];
    }
    lbo_bmv(m, sy, wt, col, &wa[2 * m], &wa[0], info);
    if (*info != 0) {
      *info = -8;
      return;
    }
    pointr = head;
    for (j = 1; j <= col; j++) {
      a1 = wa[j - 1];
      a2 = theta * wa[col + j - 1];
      for (i = 1; i <= nfree; i++) {
        k = index[i - 1];
        r[i - 1] = r[i - 1] + WY(k, pointr) * a1 + WS(k, pointr) * a2;
      }
      pointr = pointr % m + 1;
    }
  }
}

/* ---------- :1980-2059 freev ---------- */
void lbo_freev(int n, int *nfree, int *index, int *nenter, int *ileave, int *indx2,
               const int *iwhere, int *wrk, int updatd, int cnstnd, int iter) {
  int iact, i, k;
  *nenter = 0;
  *ileave = n + 1;
  if (iter > 0 && cnstnd) {
    for (i = 1; i <= *nfree; i++) {
      k = index[i - 1];
      if (iwhere[k - 1] > 0) {
        *ileave = *ileave - 1;
        indx2[*ileave - 1] = k;
      }
    }
    for (i = 1 + *nfree; i <= n; i++) {
      k = index[i - 1];
      if (iwhere[k - 1] <= 0) {
        *nenter = *nenter + 1;
        indx2[*nenter - 1] = k;
      }
    }
  }
  *wrk = (*ileave < n + 1) || (*nenter > 0) || updatd;
  *nfree = 0;
  iact = n + 1;
  for (i = 1; i <= n; i++) {
    if (iwhere[i - 1] <= 0) {
      *nfree = *nfree + 1;
      index[*nfree - 1] = i;
    } else {
      iact = iact - 1;
      index[iact - 1] = i;
    }
  }
}

/* ---------- :1681-1908 formk ---------- */
#define WN(i, j) wn[((i)-1) + (int64_t)((j)-1) * m2]
#define WN1(i, j) wn1[((i)-1) + (int64_t)((j)-1) * m2]
void lbo_formk(int n, int nsub, const int *ind, int nenter, int ileave, const int *indx2,
               int iupdat, int updatd, real *wn, real *wn1, int m, const real *ws,
               const real *wy, const real *sy, real theta, int col, int head, int *info) {
  int m2 = 2 * m, ipntr, jpntr, iy, is, jy, js, is1, js1, k1, i, k, col2, pbegin, pend, dbegin,
      dend, upcl;
  real temp1, temp2, temp3, temp4;

  if (updatd) {
    if (iupdat > m) { /* :1736-1744 shift old part of WN1 */
      for (jy = 1; jy <= m - 1; jy++) {
        js = m + jy;
        dcopy(m - jy, &WN1(jy + 1, jy + 1), &WN1(jy, jy));
        dcopy(m - jy, &WN1(js + 1, js + 1), &WN1(js, js));
        dcopy(m - 1, &WN1(m + 2, jy + 1), &WN1(m + 1, jy));
      }
    }
    pbegin = 1;
    pend = nsub;
    dbegin = nsub + 1;
    dend = n;
    iy = col;
    is = m + col;
    ipntr = head + col - 1;
    if (ipntr > m) ipntr = ipntr - m;
    jpntr = head;
    for (jy = 1; jy <= col; jy++) { /* :1756-1776 */
      js = m + jy;
      temp1 = ZERO;
      temp2 = ZERO;
      temp3 = ZERO;
      for (k = pbegin; k <= pend; k++) {
        k1 = ind[k - 1];
        temp1 = temp1 + WY(k1, ipntr) * WY(k1, jpntr);
      }
      for (k = dbegin; k <= dend; k++) {
        k1 = ind[k - 1];
        temp2 = temp2 + WS(k1, ipntr) * WS(k1, jpntr);
        temp3 = temp3 + WS(k1, ipntr) * WY(k1, jpntr);
      }
      WN1(iy, jy) = temp1;
      WN1(is, js) = temp2;
      WN1(is, jy) = temp3;
      jpntr = jpntr % m + 1;
    }
    jy = col; /* :1779-1793 */
    jpntr = head + col - 1;
    if (jpntr > m) jpntr = jpntr - m;
    ipntr = head;
    for (i = 1; i <= col; i++) {
      is = m + i;
      temp3 = ZERO;
      for (k = pbegin; k <= pend; k++) {
        k1 = ind[k - 1];
        temp3 = temp3 + WS(k1, ipntr) * WY(k1, jpntr);
      }
      ipntr = ipntr % m + 1;
      WN1(is, jy) = temp3;
    }
    upcl = col - 1;
  } else {
    upcl = col;
  }

  ipntr = head; /* :1801-1826 */
  for (iy = 1; iy <= upcl; iy++) {
    is = m + iy;
    jpntr = head;
    for (jy = 1; jy <= iy; jy++) {
      js = m + jy;
      temp1 = ZERO;
      temp2 = ZERO;
      temp3 = ZERO;
      temp4 = ZERO;
      for (k = 1; k <= nenter; k++) {
        k1 = indx2[k - 1];
        temp1 = temp1 + WY(k1, ipntr) * WY(k1, jpntr);
        temp2 = temp2 + WS(k1, ipntr) * WS(k1, jpntr);
      }
      for (k = ileave; k <= n; k++) {
        k1 = indx2[k - 1];
        temp3 = temp3 + WY(k1, ipntr) * WY(k1, jpntr);
        temp4 = temp4 + WS(k1, ipntr) * WS(k1, jpntr);
      }
      WN1(iy, jy) = WN1(iy, jy) + temp1 - temp3;
      WN1(is, js) = WN1(is, js) - temp2 + temp4;
      jpntr = jpntr % m + 1;
    }
    ipntr = ipntr % m + 1;
  }

  ipntr = head; /* :1829-1851 */
  for (is = m + 1; is <= m + upcl; is++) {
    jpntr = head;
    for (jy = 1; jy <= upcl; jy++) {
      temp1 = ZERO;
      temp3 = ZERO;
      for (k = 1; k <= nenter; k++) {
        k1 = indx2[k - 1];
        temp1 = temp1 + WS(k1, ipntr) * WY(k1, jpntr);
      }
      for (k = ileave; k <= n; k++) {
        k1 = indx2[k - 1];
        temp3 = temp3 + WS(k1, ipntr) * WY(k1, jpntr);
      }
      if (is <= jy + m)
        WN1(is, jy) = WN1(is, jy) + temp1 - temp3;
      else
        WN1(is, jy) = WN1(is, jy) - temp1 + temp3;
      jpntr = jpntr % m + 1;
    }
    ipntr = ipntr % m + 1;
  }

  for (iy = 1; iy <= col; iy++) { /* :1856-1873 */
    is = col + iy;
    is1 = m + iy;
    for (jy = 1; jy <= iy; jy++) {
      js = col + jy;
      js1 = m + jy;
      WN(jy, iy) = WN1(iy, jy) / theta;
      WN(js, is) = WN1(is1, js1) * theta;
    }
    for (jy = 1; jy <= iy - 1; jy++) WN(jy, is) = -WN1(is1, jy);
    for (jy = iy; jy <= col; jy++) WN(jy, is) = WN1(is1, jy);
    WN(iy, iy) = WN(iy, iy) + SY(iy, iy);
  }

  lbo_dpofa(wn, m2, col, info); /* :1880-1884 */
  if (*info != 0) {
    *info = -1;
    return;
  }
  col2 = 2 * col;
  for (js = col + 1; js <= col2; js++) lbo_dtrsl(wn, m2, col, &WN(1, js), 11, info);
  for (is = col + 1; is <= col2; is++)
    for (js = is; js <= col2; js++)
      WN(is, js) = WN(is, js) + lbo_ddot(col, &WN(1, is), &WN(1, js));
  lbo_dpofa(&WN(col + 1, col + 1), m2, col, info); /* :1902-1906 */
  if (*info != 0) {
    *info = -2;
    return;
  }
}

/* ---------- :1926-1963 formt ---------- */
void lbo_formt(int m, real *wt, const real *sy, const real *ss, int col, real theta,
               int *info) {
  int i, j, k, k1;
  real ddum;
  for (j = 1; j <= col; j++) WT(1, j) = theta * SS(1, j);
  for (i = 2; i <= col; i++) {
    for (j = i; j <= col; j++) {
      k1 = (i < j ? i : j) - 1;
      ddum = ZERO;
      for (k = 1; k <= k1; k++) ddum = ddum + SY(i, k) * SY(j, k) / SY(k, k);
      WT(i, j) = ddum + theta * SS(i, j);
    }
  }
  lbo_dpofa(wt, m, col, info);
  if (*info != 0) *info = -3;
}

/* ---------- :2291-2346 matupd ---------- */
void lbo_matupd(int n, int m, real *ws, real *wy, real *sy, real *ss, const real *d,
                const real *r, int *itail, int iupdat, int *col, int *head, real *theta,
                real rr, real dr, real stp, real dtd) {
  int j, pointr;
  if (iupdat <= m) {
    *col = iupdat;
    *itail = (*head + iupdat - 2) % m + 1;
  } else {
    *itail = *itail % m + 1;
    *head = *head % m + 1;
  }
  dcopy(n, d, &WS(1, *itail));
  dcopy(n, r, &WY(1, *itail));
  *theta = rr / dr;
  if (iupdat > m) {
    for (j = 1; j <= *col - 1; j++) {
      dcopy(j, &SS(2, j + 1), &SS(1, j));
      dcopy(*col - j, &SY(j + 1, j + 1), &SY(j, j));
    }
  }
  pointr = *head;
  for (j = 1; j <= *col - 1; j++) {
    SY(*col, j) = lbo_ddot(n, d, &WY(1, pointr));
    SS(j, *col) = lbo_ddot(n, &WS(1, pointr), d);
    pointr = pointr % m + 1;
  }
  if (stp == ONE)
    SS(*col, *col) = dtd;
  else
    SS(*col, *col) = stp * stp * dtd;
  SY(*col, *col) = dr;
}

/* ---------- :2676-2885 subsm ---------- */
void lbo_subsm(int n, int m, int nsub, const int *ind, const real *l, const real *u,
               const int *nbd, real *x, real *d, real *xp, const real *ws, const real *wy,
               real theta, const real *xx, const real *gg, int col, int head, int *iword,
               real *wv, const real *wn, int *info) {
  int pointr, m2, col2, ibd, jy, js, i, j, k;
  real alpha, xk, dk, temp1, temp2, dd_p;

  if (nsub <= 0) return;

  pointr = head; /* :2742-2754 */
  for (i = 1; i <= col; i++) {
    temp1 = ZERO;
    temp2 = ZERO;
    for (j = 1; j <= nsub; j++) {
      k = ind[j - 1];
      temp1 = temp1 + WY(k, pointr) * d[j - 1];
      temp2 = temp2 + WS(k, pointr) * d[j - 1];
    }
    wv[i - 1] = temp1;
    wv[col + i - 1] = theta * temp2;
    pointr = pointr % m + 1;
  }

  m2 = 2 * m; /* :2758-2766 */
  col2 = 2 * col;
  lbo_dtrsl(wn, m2, col2, wv, 11, info);
  if (*info != 0) return;
  for (i = 0; i < col; i++) wv[i] = -wv[i];
  lbo_dtrsl(wn, m2, col2, wv, 1, info);
  if (*info != 0) return;

  pointr = head; /* :2770-2780 */
  for (jy = 1; jy <= col; jy++) {
    js = col + jy;
    for (i = 1; i <= nsub; i++) {
      k = ind[i - 1];
      d[i - 1] = d[i - 1] + WY(k, pointr) * wv[jy - 1] / theta + WS(k, pointr) * wv[js - 1];
    }
    pointr = pointr % m + 1;
  }
  dscal(nsub, ONE / theta, d);

  *iword = 0; /* :2785-2816 */
  dcopy(n, x, xp);
  for (i = 1; i <= nsub; i++) {
    k = ind[i - 1];
    dk = d[i - 1];
    xk = x[k - 1];
    if (nbd[k - 1] != 0) {
      if (nbd[k - 1] == 1) {
        x[k - 1] = rmax(l[k - 1], xk + dk);
        if (x[k - 1] == l[k - 1]) *iword = 1;
      } else if (nbd[k - 1] == 2) {
        xk = rmax(l[k - 1], xk + dk);
        x[k - 1] = rmin(u[k - 1], xk);
        if (x[k - 1] == l[k - 1] || x[k - 1] == u[k - 1]) *iword = 1;
      } else if (nbd[k - 1] == 3) {
        x[k - 1] = rmin(u[k - 1], xk + dk);
        if (x[k - 1] == u[k - 1]) *iword = 1;
      }
    } else {
      x[k - 1] = xk + dk;
    }
  }

  if (*iword == 0) return; /* :2820 */
  dd_p = ZERO;             /* :2824-2828 */
  for (i = 0; i < n; i++) dd_p = dd_p + (x[i] - xx[i]) * gg[i];
  if (dd_p <= ZERO) return;

  dcopy(n, xp, x); /* :2830-2879 */
  lbo_branch_count[0]++;
  alpha = ONE;
  temp1 = alpha;
  ibd = 0;
  for (i = 1; i <= nsub; i++) {
    k = ind[i - 1];
    dk = d[i - 1];
    if (nbd[k - 1] != 0) {
      if (dk < ZERO && nbd[k - 1] <= 2) {
        temp2 = l[k - 1] - x[k - 1];
        if (temp2 >= ZERO)
          temp1 = ZERO;
        else if (dk * alpha < temp2)
          temp1 = temp2 / dk;
      } else if (dk > ZERO && nbd[k - 1] >= 2) {
        temp2 = u[k - 1] - x[k - 1];
        if (temp2 <= ZERO)
          temp1 = ZERO;
        else if (dk * alpha > temp2)
          temp1 = temp2 / dk;
      }
      if (temp1 < alpha) {
        alpha = temp1;
        ibd = i;
      }
    }
  }
  if (alpha < ONE) {
    dk = d[ibd - 1];
    k = ind[ibd - 1];
    if (dk > ZERO) {
      x[k - 1] = u[k - 1];
      d[ibd - 1] = ZERO;
    } else if (dk < ZERO) {
      x[k - 1] = l[k - 1];
      d[ibd - 1] = ZERO;
    }
  }
  for (i = 1; i <= nsub; i++) {
    k = ind[i - 1];
    x[k - 1] = x[k - 1] + alpha * d[i - 1];
  }
}

/* ---------- :3227-3415 dcstep ---------- */
static void dcstep(real *stx, real *fx, real *dx, real *sty, real *fy, real *dy, real *stp,
                   real fp, real dp, int *brackt, real stpmin, real stpmax) {
  const real p66 = (real)0.66;
  real gamma, p, q, r, s, sgnd, stpc, stpf, stpq, theta, tmp;

  sgnd = dp * (*dx / RABS(*dx));
  if (fp > *fx) {
    theta = THREE * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = rmax(rmax(RABS(theta), RABS(*dx)), RABS(dp));
    tmp = theta / s;
    gamma = s * RSQRT(tmp * tmp - (*dx / s) * (dp / s));
    if (*stp < *stx) gamma = -gamma;
    p = (gamma - *dx) + theta;
    q = ((gamma - *dx) + gamma) + dp;
    r = p / q;
    stpc = *stx + r * (*stp - *stx);
    stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / TWO) * (*stp - *stx);
    if (RABS(stpc - *stx) < RABS(stpq - *stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / TWO;
    *brackt = 1;
  } else if (sgnd < ZERO) {
    theta = THREE * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = rmax(rmax(RABS(theta), RABS(*dx)), RABS(dp));
    tmp = theta / s;
    gamma = s * RSQRT(tmp * tmp - (*dx / s) * (dp / s));
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + *dx;
    r = p / q;
    stpc = *stp + r * (*stx - *stp);
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (RABS(stpc - *stp) > RABS(stpq - *stp))
      stpf = stpc;
    else
      stpf = stpq;
    *brackt = 1;
  } else if (RABS(dp) < RABS(*dx)) {
    theta = THREE * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = rmax(rmax(RABS(theta), RABS(*dx)), RABS(dp));
    tmp = theta / s;
    gamma = s * RSQRT(rmax(ZERO, tmp * tmp - (*dx / s) * (dp / s)));
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (*dx - dp)) + gamma;
    r = p / q;
    if (r < ZERO && gamma != ZERO)
      stpc = *stp + r * (*stx - *stp);
    else if (*stp > *stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (*brackt) {
      if (RABS(stpc - *stp) < RABS(stpq - *stp))
        stpf = stpc;
      else
        stpf = stpq;
      if (*stp > *stx)
        stpf = rmin(*stp + p66 * (*sty - *stp), stpf);
      else
        stpf = rmax(*stp + p66 * (*sty - *stp), stpf);
    } else {
      if (RABS(stpc - *stp) > RABS(stpq - *stp))
        stpf = stpc;
      else
        stpf = stpq;
      stpf = rmin(stpmax, stpf);
      stpf = rmax(stpmin, stpf);
    }
  } else {
    if (*brackt) {
      theta = THREE * (fp - *fy) / (*sty - *stp) + *dy + dp;
      s = rmax(rmax(RABS(theta), RABS(*dy)), RABS(dp));
      tmp = theta / s;
      gamma = s * RSQRT(tmp * tmp - (*dy / s) * (dp / s));
      if (*stp > *sty) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + *dy;
      r = p / q;
      stpc = *stp + r * (*sty - *stp);
      stpf = stpc;
    } else if (*stp > *stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > *fx) {
    *sty = *stp;
    *fy = fp;
    *dy = dp;
  } else {
    if (sgnd < ZERO) {
      *sty = *stx;
      *fy = *fx;
      *dy = *dx;
    }
    *stx = *stp;
    *fx = fp;
    *dx = dp;
  }
  *stp = stpf;
}

/* ---------- :2942-3198 dcsrch ---------- */
void lbo_dcsrch(real *f, real *g, real *stp, real ftol, real gtol, real xtol, real stpmin,
                real stpmax, char *task, int *isave, real *dsave) {
  const real p5 = (real)0.5, p66 = (real)0.66, xtrapl = (real)1.1, xtrapu = (real)4.0;
  int brackt, stage;
  real finit, ftest, fm, fx, fxm, fy, fym, ginit, gtest, gm, gx, gxm, gy, gym, stx, sty, stmin,
      stmax, width, width1;

  if (task_pre(task, "START")) {
    if (*stp < stpmin) task_set(task, "ERROR: STP < STPMIN");
    if (*stp > stpmax) task_set(task, "ERROR: STP > STPMAX");
    if (*g >= ZERO) task_set(task, "ERROR: INITIAL G >= ZERO");
    if (ftol < ZERO) task_set(task, "ERROR: FTOL < ZERO");
    if (gtol < ZERO) task_set(task, "ERROR: GTOL < ZERO");
    if (xtol < ZERO) task_set(task, "ERROR: XTOL < ZERO");
    if (stpmin < ZERO) task_set(task, "ERROR: STPMIN < ZERO");
    if (stpmax < stpmin) task_set(task, "ERROR: STPMAX < STPMIN");
    if (task_pre(task, "ERROR")) return;
    brackt = 0;
    stage = 1;
    finit = *f;
    ginit = *g;
    gtest = ftol * ginit;
    width = stpmax - stpmin;
    width1 = width / p5;
    stx = ZERO;
    fx = finit;
    gx = ginit;
    sty = ZERO;
    fy = finit;
    gy = ginit;
    stmin = ZERO;
    stmax = *stp + xtrapu * *stp;
    task_set(task, "FG");
    goto save;
  } else {
    brackt = isave[0] == 1;
    stage = isave[1];
    ginit = dsave[0];
    gtest = dsave[1];
    gx = dsave[2];
    gy = dsave[3];
    finit = dsave[4];
    fx = dsave[5];
    fy = dsave[6];
    stx = dsave[7];
    sty = dsave[8];
    stmin = dsave[9];
    stmax = dsave[10];
    width = dsave[11];
    width1 = dsave[12];
  }

  ftest = finit + *stp * gtest;
  if (stage == 1 && *f <= ftest && *g >= ZERO) stage = 2;

  if (brackt && (*stp <= stmin || *stp >= stmax))
    task_set(task, "WARNING: ROUNDING ERRORS PREVENT PROGRESS");
  if (brackt && stmax - stmin <= xtol * stmax) task_set(task, "WARNING: XTOL TEST SATISFIED");
  if (*stp == stpmax && *f <= ftest && *g <= gtest) task_set(task, "WARNING: STP = STPMAX");
  if (*stp == stpmin && (*f > ftest || *g >= gtest)) task_set(task, "WARNING: STP = STPMIN");

  if (*f <= ftest && RABS(*g) <= gtol * (-ginit)) task_set(task, "CONVERGENCE");

  if (task_pre(task, "WARN") || task_pre(task, "CONV")) goto save;

  if (stage == 1 && *f <= fx && *f > ftest) {
    fm = *f - *stp * gtest;
    fxm = fx - stx * gtest;
    fym = fy - sty * gtest;
    gm = *g - gtest;
    gxm = gx - gtest;
    gym = gy - gtest;
    dcstep(&stx, &fxm, &gxm, &sty, &fym, &gym, stp, fm, gm, &brackt, stmin, stmax);
    fx = fxm + stx * gtest;
    fy = fym + sty * gtest;
    gx = gxm + gtest;
    gy = gym + gtest;
  } else {
    dcstep(&stx, &fx, &gx, &sty, &fy, &gy, stp, *f, *g, &brackt, stmin, stmax);
  }

  if (brackt) {
    if (RABS(sty - stx) >= p66 * width1) *stp = stx + p5 * (sty - stx);
    width1 = width;
    width = RABS(sty - stx);
  }
  if (brackt) {
    stmin = rmin(stx, sty);
    stmax = rmax(stx, sty);
  } else {
    stmin = *stp + xtrapl * (*stp - stx);
    stmax = *stp + xtrapu * (*stp - stx);
  }
  *stp = rmax(*stp, stpmin);
  *stp = rmin(*stp, stpmax);
  if ((brackt && (*stp <= stmin || *stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax))
    *stp = stx;
  task_set(task, "FG");

save:
  isave[0] = brackt ? 1 : 0;
  isave[1] = stage;
  dsave[0] = ginit;
  dsave[1] = gtest;
  dsave[2] = gx;
  dsave[3] = gy;
  dsave[4] = finit;
  dsave[5] = fx;
  dsave[6] = fy;
  dsave[7] = stx;
  dsave[8] = sty;
  dsave[9] = stmin;
  dsave[10] = stmax;
  dsave[11] = width;
  dsave[12] = width1;
}

/* ---------- :2174-2275 lnsrlb ---------- */
void lbo_lnsrlb(int n, const real *l, const real *u, const int *nbd, real *x, real f,
                real *fold, real *gd, real *gdold, const real *g, const real *d, real *r,
                real *t, const real *z, real *stp, real *dnorm, real *dtd, real *xstep,
                real *stpmx, int iter, int *ifun, int *iback, int *nfgv, int *info, char *task,
                int boxed, int cnstnd, char *csave, int *isave2, real *dsave13) {
  const real big = (real)1.0e+10, ftol = (real)1.0e-3, gtol = (real)0.9, xtol = (real)0.1;
  int i;
  real a1, a2, fcopy, gdcopy;

  if (!task_pre(task, "FG_LN")) {
    *dtd = lbo_ddot(n, d, d);
    *dnorm = RSQRT(*dtd);
    *stpmx = big;
    if (cnstnd) {
      if (iter == 0) {
        *stpmx = ONE;
      } else {
        for (i = 0; i < n; i++) {
          a1 = d[i];
          if (nbd[i] != 0) {
            if (a1 < ZERO && nbd[i] <= 2) {
              a2 = l[i] - x[i];
              if (a2 >= ZERO)
                *stpmx = ZERO;
              else if (a1 * *stpmx < a2)
                *stpmx = a2 / a1;
            } else if (a1 > ZERO && nbd[i] >= 2) {
              a2 = u[i] - x[i];
              if (a2 <= ZERO)
                *stpmx = ZERO;
              else if (a1 * *stpmx > a2)
                *stpmx = a2 / a1;
            }
          }
        }
      }
    }
    if (iter == 0 && !boxed)
      *stp = rmin(ONE / *dnorm, *stpmx);
    else
      *stp = ONE;
    dcopy(n, x, t);
    dcopy(n, g, r);
    *fold = f;
    *ifun = 0;
    *iback = 0;
    task_set(csave, "START");
  }

  *gd = lbo_ddot(n, g, d);
  if (*ifun == 0) {
    *gdold = *gd;
    if (*gd >= ZERO) {
      *info = -4; /* ' ascent direction in projection gd = ' (:2250) not printed */
      return;
    }
  }
  fcopy = f;
  gdcopy = *gd;
  lbo_dcsrch(&fcopy, &gdcopy, stp, ftol, gtol, xtol, ZERO, *stpmx, csave, isave2, dsave13);
  *xstep = *stp * *dnorm;
  if (!task_pre(csave, "CONV") && !task_pre(csave, "WARN")) {
    task_set(task, "FG_LNSRCH");
    *ifun = *ifun + 1;
    *nfgv = *nfgv + 1;
    *iback = *ifun - 1;
    if (*stp == ONE) {
      dcopy(n, z, x);
    } else {
      for (i = 0; i < n; i++) x[i] = *stp * d[i] + t[i];
    }
  } else {
    task_set(task, "NEW_X");
  }
}

/* ---------- :312-949 mainlb ---------- */
static void mainlb(int n, int m, real *x, const real *l, const real *u, const int *nbd, real *f,
                   real *g, real factr, real pgtol, real *ws, real *wy, real *sy, real *ss,
                   real *wt, real *wn, real *snd, real *z, real *r, real *d, real *t, real *xp,
                   real *wa, int *index, int *iwhere, int *indx2, char *task, int iprint,
                   char *csave, int *lsave, int *isave, real *dsave) {
  int prjctd, cnstnd, boxed, updatd, wrk = 0;
  int i, k = 0, nintol, itfile = 0, iback, nskip, head, col, iter, itail, iupdat, nseg, nfgv,
         info, ifun, iword, nfree, nact, ileave, nenter;
  real theta, fold, dr, rr, tol, xstep = ZERO, sbgnrm, ddum, dnorm, dtd, epsmch, cpu1, cpu2,
                                cachyt, sbtime, lnscht, time1, gd, gdold, stp, stpmx;
  int compute_pg, prelims, linesearch;
  (void)iprint;
  (void)k;

#define SAVE_LOCALS()                                                                           \
  do {                                                                                          \
    lsave[0] = prjctd;                                                                          \
    lsave[1] = cnstnd;                                                                          \
    lsave[2] = boxed;                                                                           \
    lsave[3] = updatd;                                                                          \
    isave[0] = nintol;                                                                          \
    isave[2] = itfile;                                                                          \
    isave[3] = iback;                                                                           \
    isave[4] = nskip;                                                                           \
    isave[5] = head;                                                                            \
    isave[6] = col;                                                                             \
    isave[7] = itail;                                                                           \
    isave[8] = iter;                                                                            \
    isave[9] = iupdat;                                                                          \
    isave[11] = nseg;                                                                           \
    isave[12] = nfgv;                                                                           \
    isave[13] = info;                                                                           \
    isave[14] = ifun;                                                                           \
    isave[15] = iword;                                                                          \
    isave[16] = nfree;                                                                          \
    isave[17] = nact;                                                                           \
    isave[18] = ileave;                                                                         \
    isave[19] = nenter;                                                                         \
    dsave[0] = theta;                                                                           \
    dsave[1] = fold;                                                                            \
    dsave[2] = tol;                                                                             \
    dsave[3] = dnorm;                                                                           \
    dsave[4] = epsmch;                                                                          \
    dsave[5] = cpu1;                                                                            \
    dsave[6] = cachyt;                                                                          \
    dsave[7] = sbtime;                                                                          \
    dsave[8] = lnscht;                                                                          \
    dsave[9] = time1;                                                                           \
    dsave[10] = gd;                                                                             \
    dsave[11] = stpmx;                                                                          \
    dsave[12] = sbgnrm;                                                                         \
    dsave[13] = stp;                                                                            \
    dsave[14] = gdold;                                                                          \
    dsave[15] = dtd;                                                                            \
  } while (0)

#define REFRESH_MEMORY()                                                                        \
  do {                                                                                          \
    lbo_branch_count[1]++;                                                                      \
    info = 0;                                                                                   \
    col = 0;                                                                                    \
    head = 1;                                                                                   \
    theta = ONE;                                                                                \
    iupdat = 0;                                                                                 \
    updatd = 0;                                                                                 \
  } while (0)

  if (task_eq(task, "START")) { /* :430-507 */
    epsmch = REPS;
    time1 = (real)cpu_now();
    col = 0;
    head = 1;
    theta = ONE;
    iupdat = 0;
    updatd = 0;
    iback = 0;
    itail = 0;
    iword = 0;
    nact = 0;
    ileave = 0;
    nenter = 0;
    fold = ZERO;
    dnorm = ZERO;
    cpu1 = ZERO;
    gd = ZERO;
    stpmx = ZERO;
    sbgnrm = ZERO;
    stp = ZERO;
    gdold = ZERO;
    dtd = ZERO;
    iter = 0;
    nfgv = 0;
    nseg = 0;
    nintol = 0;
    nskip = 0;
    nfree = n;
    ifun = 0;
    tol = factr * epsmch;
    cachyt = 0;
    sbtime = 0;
    lnscht = 0;
    info = 0;
    lbo_errclb(n, m, factr, l, u, nbd, task, &info, &k);
    if (task_pre(task, "ERROR")) return; /* prn3lb only, no save_locals (:492-497) */
    lbo_active(n, l, u, nbd, x, iwhere, &prjctd, &cnstnd, &boxed, 0);
    task_set(task, "FG_START");
    SAVE_LOCALS();
    return;
  }

  /* :511-550 restore */
  prjctd = lsave[0];
  cnstnd = lsave[1];
  boxed = lsave[2];
  updatd = lsave[3];
  nintol = isave[0];
  itfile = isave[2];
  iback = isave[3];
  nskip = isave[4];
  head = isave[5];
  col = isave[6];
  itail = isave[7];
  iter = isave[8];
  iupdat = isave[9];
  nseg = isave[11];
  nfgv = isave[12];
  info = isave[13];
  ifun = isave[14];
  iword = isave[15];
  nfree = isave[16];
  nact = isave[17];
  ileave = isave[18];
  nenter = isave[19];
  theta = dsave[0];
  fold = dsave[1];
  tol = dsave[2];
  dnorm = dsave[3];
  epsmch = dsave[4];
  cpu1 = dsave[5];
  cachyt = dsave[6];
  sbtime = dsave[7];
  lnscht = dsave[8];
  time1 = dsave[9];
  gd = dsave[10];
  stpmx = dsave[11];
  sbgnrm = dsave[12];
  stp = dsave[13];
  gdold = dsave[14];
  dtd = dsave[15];

  compute_pg = 1; /* :554-577 */
  prelims = 1;
  linesearch = 1;
  if (task_pre(task, "FG_LN")) {
    compute_pg = 0;
    prelims = 0;
  } else if (task_pre(task, "NEW_X")) {
    compute_pg = 0;
    prelims = 0;
    linesearch = 0;
  } else if (!task_pre(task, "FG_ST")) {
    if (task_pre(task, "STOP")) {
      if (strncmp(task + 6, "CPU", 3) == 0) {
        dcopy(n, t, x);
        dcopy(n, r, g);
        *f = fold;
      }
      SAVE_LOCALS(); /* finish() */
    } else {
      task_set(task, "FG_START"); /* start() */
      SAVE_LOCALS();
    }
    return;
  }

  if (compute_pg) { /* :579-596 */
    nfgv = 1;
    sbgnrm = lbo_projgr(n, l, u, nbd, x, g);
    if (sbgnrm <= pgtol) {
      task_set(task, "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL");
      SAVE_LOCALS();
      return;
    }
  }

  for (;;) { /* main_loop :599 */
    if (prelims) {
      iword = -1;
      if (!cnstnd && col > 0) { /* :607-611 */
        dcopy(n, x, z);
        wrk = updatd;
        nseg = 0;
      } else {
        cpu1 = (real)cpu_now();
        lbo_cauchy(n, x, l, u, nbd, g, indx2, iwhere, t, d, z, m, wy, ws, sy, wt, theta, col,
                   head, &wa[0], &wa[2 * m], &wa[4 * m], &wa[6 * m], &nseg, sbgnrm, &info,
                   epsmch);
        if (info != 0) { /* :620-635 */
          REFRESH_MEMORY();
          cpu2 = (real)cpu_now();
          cachyt = cachyt + cpu2 - cpu1;
          prelims = 1;
          linesearch = 1;
          continue;
        }
        cpu2 = (real)cpu_now();
        cachyt = cachyt + cpu2 - cpu1;
        nintol = nintol + nseg;
        lbo_freev(n, &nfree, index, &nenter, &ileave, indx2, iwhere, &wrk, updatd, cnstnd,
                  iter);
        nact = n - nfree;
      }

      if (nfree == 0 || col == 0) {
        /* skip the subspace minimization :648-651 */
      } else {
        cpu1 = (real)cpu_now();
        if (wrk)
          lbo_formk(n, nfree, index, nenter, ileave, indx2, iupdat, updatd, wn, snd, m, ws, wy,
                    sy, theta, col, head, &info);
        if (info != 0) { /* :666-682 */
          REFRESH_MEMORY();
          cpu2 = (real)cpu_now();
          sbtime = sbtime + cpu2 - cpu1;
          prelims = 1;
          linesearch = 1;
          continue;
        }
        lbo_cmprlb(n, m, x, g, ws, wy, sy, wt, z, r, wa, index, theta, col, head, nfree,
                   cnstnd, &info);
        if (info == 0)
          lbo_subsm(n, m, nfree, index, l, u, nbd, z, r, xp, ws, wy, theta, x, g, col, head,
                    &iword, wa, wn, &info);
        if (info != 0) { /* :694-710 */
          REFRESH_MEMORY();
          cpu2 = (real)cpu_now();
          sbtime = sbtime + cpu2 - cpu1;
          prelims = 1;
          linesearch = 1;
          continue;
        }
        cpu2 = (real)cpu_now();
        sbtime = sbtime + cpu2 - cpu1;
      }

      for (i = 0; i < n; i++) d[i] = z[i] - x[i]; /* :720-722 */
      cpu1 = (real)cpu_now();
    }

    if (linesearch) { /* :729-790 */
      lbo_lnsrlb(n, l, u, nbd, x, *f, &fold, &gd, &gdold, g, d, r, t, z, &stp, &dnorm, &dtd,
                 &xstep, &stpmx, iter, &ifun, &iback, &nfgv, &info, task, boxed, cnstnd, csave,
                 &isave[21], &dsave[16]);
      if (info != 0 || iback >= 20) {
        dcopy(n, t, x);
        dcopy(n, r, g);
        *f = fold;
        if (col == 0) {
          if (info == 0) {
            info = -9;
            nfgv = nfgv - 1;
            ifun = ifun - 1;
            iback = iback - 1;
          }
          task_set(task, "ABNORMAL_TERMINATION_IN_LNSRCH");
          iter = iter + 1;
          SAVE_LOCALS();
          return;
        } else {
          if (info == 0) nfgv = nfgv - 1;
          REFRESH_MEMORY();
          task_set(task, "RESTART_FROM_LNSRCH");
          cpu2 = (real)cpu_now();
          lnscht = lnscht + cpu2 - cpu1;
          prelims = 1;
          linesearch = 1;
          continue;
        }
      } else if (task_pre(task, "FG_LN")) {
        SAVE_LOCALS();
        return;
      } else {
        cpu2 = (real)cpu_now();
        lnscht = lnscht + cpu2 - cpu1;
        iter = iter + 1;
        sbgnrm = lbo_projgr(n, l, u, nbd, x, g);
        SAVE_LOCALS();
        return;
      }
    }

    /* :794-810 termination tests */
    if (sbgnrm <= pgtol) {
      task_set(task, "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL");
      SAVE_LOCALS();
      return;
    }
    ddum = rmax(rmax(RABS(fold), RABS(*f)), ONE);
    if ((fold - *f) <= tol * ddum) {
      task_set(task, "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH");
      if (iback >= 10) info = -5;
      SAVE_LOCALS();
      return;
    }

    /* :812-834 */
    for (i = 0; i < n; i++) r[i] = g[i] - r[i];
    rr = lbo_ddot(n, r, r);
    if (stp == ONE) {
      dr = gd - gdold;
      ddum = -gdold;
    } else {
      dr = (gd - gdold) * stp;
      dscal(n, stp, d);
      ddum = -gdold * stp;
    }
    if (dr <= epsmch * ddum) {
      nskip = nskip + 1;
      updatd = 0;
      prelims = 1;
      linesearch = 1;
      continue;
    }

    /* :836-870 */
    updatd = 1;
    iupdat = iupdat + 1;
    lbo_matupd(n, m, ws, wy, sy, ss, d, r, &itail, iupdat, &col, &head, &theta, rr, dr, stp,
               dtd);
    lbo_formt(m, wt, sy, ss, col, theta, &info);
    if (info != 0) REFRESH_MEMORY();
    prelims = 1;
    linesearch = 1;
  }
#undef SAVE_LOCALS
#undef REFRESH_MEMORY
}

/* ---------- :88-286 setulb ---------- */
void lbo_setulb(int n, int m, real *x, const real *l, const real *u, const int *nbd, real *f,
                real *g, real factr, real pgtol, real *wa, int *iwa, char *task, int iprint,
                char *csave, int *lsave, int *isave, real *dsave) {
  int64_t off[13];
  lbo_wa_offsets(n, m, off);
  if (task_eq(task, "START")) {
    int i;
    /* :250-265; the reference stores these in default integers (they wrap
     * at n=1e8); offsets actually used below are the 64-bit off[]. */
    isave[0] = (int)((int64_t)m * n);
    isave[1] = m * m;
    isave[2] = 4 * m * m;
    for (i = 0; i < 13; i++) isave[3 + i] = (int)(off[i] + 1);
  }
  mainlb(n, m, x, l, u, nbd, f, g, factr, pgtol, wa + off[0], wa + off[1], wa + off[2],
         wa + off[3], wa + off[4], wa + off[5], wa + off[6], wa + off[7], wa + off[8],
         wa + off[9], wa + off[10], wa + off[11], wa + off[12], iwa, iwa + n,
         iwa + 2 * (int64_t)n, task, iprint, csave, lsave, isave + 21, dsave);
}

/* ---------- synthetic objectives (SURVEY.md 8d / BASELINE.md 3) ---------- */
real lbo_quadratic_fg(int64_t n, int64_t i0, const real *x, real *g) {
  /* a_i = 1 + 99*mod(7919 i,10007)/10006 ; c_i = -2 + 4*mod(104729 i,100003)/100002
   * (1-based global i, int64 products); f = 1/2 sum a (x-c)^2 ; g = a (x-c).
   * Sequential sum in index order. */
  int64_t k;
  real fsum = ZERO;
  for (k = 0; k < n; k++) {
    int64_t i = i0 + k + 1;
    real a = (real)1 + (real)99 * (real)((7919 * i) % 10007) / (real)10006;
    real c = (real)-2 + (real)4 * (real)((104729 * i) % 100003) / (real)100002;
    real dx = x[k] - c;
    g[k] = a * dx;
    fsum = fsum + a * dx * dx;
  }
  return (real)0.5 * fsum;
}

/* test/driver1.f90:274-289 (same formulas in driver2/driver3) */
real lbo_rosenbrock_fg(int64_t n, const real *x, real *g) {
  int64_t i;
  real f, t1, t2, tmp;
  tmp = x[0] - ONE;
  f = (real)0.25 * (tmp * tmp);
  for (i = 1; i < n; i++) {
    tmp = x[i] - x[i - 1] * x[i - 1];
    f = f + tmp * tmp;
  }
  f = (real)4 * f;
  t1 = x[1] - x[0] * x[0];
  g[0] = TWO * (x[0] - ONE) - (real)16 * x[0] * t1;
  for (i = 1; i < n - 1; i++) {
    t2 = t1;
    t1 = x[i + 1] - x[i] * x[i];
    g[i] = (real)8 * t2 - (real)16 * x[i] * t1;
  }
  g[n - 1] = (real)8 * t1;
  return f;
}
