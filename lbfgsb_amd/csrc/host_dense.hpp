// host_dense.hpp -- the 2m x 2m algebra of the L-BFGS-B iteration, on the host.
//
// BASELINE.json north_star: "the 2m x 2m bmv middle-matrix solve and dpofa/dtrsl
// from lbfgsb_linpack_module stay on the host".  Everything here is O(m^2) or
// O(m^3) scalar work on matrices of order <= 2m <= 64; the n-dimensional work is
// in the k_*.hip files.  Arithmetic is fp64 for both REAL64 and REAL32 contexts.
//
// Follows (operation order included, so the CPU oracle can be compared to the
// last bit on identical inputs):
//   dpofa   reference src/lbfgsb_linpack_module.f90:30-67
//   dtrsl   reference src/lbfgsb_linpack_module.f90:87-165
//   bmv     reference src/lbfgsb.f90:1057-1123
//   formt   reference src/lbfgsb.f90:1926-1963
//   dcsrch  reference src/lbfgsb.f90:2942-3198
//   dcstep  reference src/lbfgsb.f90:3227-3415
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace lbh {

// column-major view, 0-based
struct Mat {
  double *p;
  int ld;
  double &operator()(int i, int j) const { return p[i + (size_t)j * ld]; }
};

inline double dot_seq(int n, const double *a, const double *b) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s = s + a[i] * b[i];
  return s;
}

// Cholesky A = R'R in the upper triangle; returns 0 or the order of the
// leading minor that is not positive definite.
inline int dpofa(Mat a, int n) {
  for (int j = 0; j < n; ++j) {
    double s = 0.0;
    for (int k = 0; k < j; ++k) {
      double t = a(k, j) - dot_seq(k, &a(0, k), &a(0, j));
      t = t / a(k, k);
      a(k, j) = t;
      s = s + t * t;
    }
    s = a(j, j) - s;
    if (s <= 0.0) return j + 1;
    a(j, j) = std::sqrt(s);
  }
  return 0;
}

// Triangular solves; job 00: T x = b (T lower), 01: T x = b (T upper),
// 10: T' x = b (T lower), 11: T' x = b (T upper).  Returns 0 or the 1-based
// index of a zero diagonal element.
inline int dtrsl(Mat t, int n, double *b, int job) {
  for (int i = 0; i < n; ++i)
    if (t(i, i) == 0.0) return i + 1;
  int kase = 1;
  if (job % 10 != 0) kase = 2;
  if ((job % 100) / 10 != 0) kase += 2;
  switch (kase) {
    case 1:
      b[0] = b[0] / t(0, 0);
      for (int j = 1; j < n; ++j) {
        double temp = -b[j - 1];
        if (temp != 0.0)
          for (int i = j; i < n; ++i) b[i] = b[i] + temp * t(i, j - 1);
        b[j] = b[j] / t(j, j);
      }
      break;
    case 2:
      b[n - 1] = b[n - 1] / t(n - 1, n - 1);
      for (int j = n - 2; j >= 0; --j) {
        double temp = -b[j + 1];
        if (temp != 0.0)
          for (int i = 0; i <= j; ++i) b[i] = b[i] + temp * t(i, j + 1);
        b[j] = b[j] / t(j, j);
      }
      break;
    case 3:
      b[n - 1] = b[n - 1] / t(n - 1, n - 1);
      for (int j = n - 2; j >= 0; --j) {
        b[j] = b[j] - dot_seq(n - 1 - j, &t(j + 1, j), &b[j + 1]);
        b[j] = b[j] / t(j, j);
      }
      break;
    case 4:
      b[0] = b[0] / t(0, 0);
      for (int j = 1; j < n; ++j) {
        b[j] = b[j] - dot_seq(j, &t(0, j), &b[0]);
        b[j] = b[j] / t(j, j);
      }
      break;
  }
  return 0;
}

// p = M v, M the 2col x 2col middle matrix of the compact L-BFGS formula.
inline int bmv(int m, const double *sy_, const double *wt_, int col, const double *v,
               double *p) {
  if (col == 0) return 0;
  Mat sy{const_cast<double *>(sy_), m}, wt{const_cast<double *>(wt_), m};
  p[col] = v[col];
  for (int i = 1; i < col; ++i) {
    double sum = 0.0;
    for (int k = 0; k < i; ++k) sum = sum + sy(i, k) * v[k] / sy(k, k);
    p[col + i] = v[col + i] + sum;
  }
  int info = dtrsl(wt, col, p + col, 11);
  if (info) return info;
  for (int i = 0; i < col; ++i) p[i] = v[i] / std::sqrt(sy(i, i));
  info = dtrsl(wt, col, p + col, 1);
  if (info) return info;
  for (int i = 0; i < col; ++i) p[i] = -p[i] / std::sqrt(sy(i, i));
  for (int i = 0; i < col; ++i) {
    double sum = 0.0;
    for (int k = i + 1; k < col; ++k) sum = sum + sy(k, i) * p[col + k] / sy(i, i);
    p[i] = p[i] + sum;
  }
  return 0;
}

// T = theta*S'S + L D^-1 L' (upper), then Cholesky.  Returns 0 or -3.
inline int formt(int m, double *wt_, const double *sy_, const double *ss_, int col,
                 double theta) {
  Mat wt{wt_, m}, sy{const_cast<double *>(sy_), m}, ss{const_cast<double *>(ss_), m};
  for (int j = 0; j < col; ++j) wt(0, j) = theta * ss(0, j);
  for (int i = 1; i < col; ++i)
    for (int j = i; j < col; ++j) {
      int k1 = std::min(i, j);
      double ddum = 0.0;
      for (int k = 0; k < k1; ++k) ddum = ddum + sy(i, k) * sy(j, k) / sy(k, k);
      wt(i, j) = ddum + theta * ss(i, j);
    }
  return dpofa(wt, col) ? -3 : 0;
}

// ---- 60-byte blank padded strings (Fortran character(len=60)) ----
inline void str60_set(char *t, const char *s) {
  size_t k = std::strlen(s);
  if (k > 60) k = 60;
  std::memcpy(t, s, k);
  std::memset(t + k, ' ', 60 - k);
}
inline bool str60_pre(const char *t, const char *s) { return std::strncmp(t, s, std::strlen(s)) == 0; }
inline bool str60_eq(const char *t, const char *s) {
  size_t k = std::strlen(s);
  if (std::strncmp(t, s, k) != 0) return false;
  for (size_t i = k; i < 60; ++i)
    if (t[i] != ' ') return false;
  return true;
}

// ---- More'-Thuente line search (MINPACK-2) ----
inline void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy,
                   double &stp, double fp, double dp, bool &brackt, double stpmin,
                   double stpmax) {
  const double p66 = 0.66;
  double gamma, p, q, r, s, stpc, stpf, stpq, theta;
  const double sgnd = dp * (dx / std::fabs(dx));
  auto max3 = [](double a, double b, double c) { return std::max(std::max(a, b), c); };
  auto sq = [](double a) { return a * a; };

  if (fp > fx) {
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = max3(std::fabs(theta), std::fabs(dx), std::fabs(dp));
    gamma = s * std::sqrt(sq(theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    p = (gamma - dx) + theta;
    q = ((gamma - dx) + gamma) + dp;
    r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    stpf = std::fabs(stpc - stx) < std::fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = max3(std::fabs(theta), std::fabs(dx), std::fabs(dp));
    gamma = s * std::sqrt(sq(theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + dx;
    r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
    brackt = true;
  } else if (std::fabs(dp) < std::fabs(dx)) {
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = max3(std::fabs(theta), std::fabs(dx), std::fabs(dp));
    gamma = s * std::sqrt(std::max(0.0, sq(theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (dx - dp)) + gamma;
    r = p / q;
    if (r < 0.0 && gamma != 0.0)
      stpc = stp + r * (stx - stp);
    else if (stp > stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      stpf = std::fabs(stpc - stp) < std::fabs(stpq - stp) ? stpc : stpq;
      if (stp > stx)
        stpf = std::min(stp + p66 * (sty - stp), stpf);
      else
        stpf = std::max(stp + p66 * (sty - stp), stpf);
    } else {
      stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
      stpf = std::min(stpmax, stpf);
      stpf = std::max(stpmin, stpf);
    }
  } else {
    if (brackt) {
      theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      s = max3(std::fabs(theta), std::fabs(dy), std::fabs(dp));
      gamma = s * std::sqrt(sq(theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + dy;
      r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stp = stpf;
}

// State lives in the caller's isave(43:44) / dsave(17:29) exactly as in the
// reference, so a reference-style caller can inspect or checkpoint it.
inline void dcsrch(double f, double g, double &stp, double ftol, double gtol, double xtol,
                   double stpmin, double stpmax, char *task, int32_t *isave, double *dsave) {
  const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
  bool brackt;
  int stage;
  double finit, ftest, fx, fy, ginit, gtest, gx, gy, stx, sty, stmin, stmax, width, width1;

  auto save = [&]() {
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
  };

  if (str60_pre(task, "START")) {
    if (stp < stpmin) str60_set(task, "ERROR: STP < STPMIN");
    if (stp > stpmax) str60_set(task, "ERROR: STP > STPMAX");
    if (g >= 0.0) str60_set(task, "ERROR: INITIAL G >= ZERO");
    if (ftol < 0.0) str60_set(task, "ERROR: FTOL < ZERO");
    if (gtol < 0.0) str60_set(task, "ERROR: GTOL < ZERO");
    if (xtol < 0.0) str60_set(task, "ERROR: XTOL < ZERO");
    if (stpmin < 0.0) str60_set(task, "ERROR: STPMIN < ZERO");
    if (stpmax < stpmin) str60_set(task, "ERROR: STPMAX < STPMIN");
    if (str60_pre(task, "ERROR")) return;
    brackt = false;
    stage = 1;
    finit = f;
    ginit = g;
    gtest = ftol * ginit;
    width = stpmax - stpmin;
    width1 = width / p5;
    stx = 0.0;
    fx = finit;
    gx = ginit;
    sty = 0.0;
    fy = finit;
    gy = ginit;
    stmin = 0.0;
    stmax = stp + xtrapu * stp;
    str60_set(task, "FG");
    save();
    return;
  }
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

  ftest = finit + stp * gtest;
  if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;

  if (brackt && (stp <= stmin || stp >= stmax))
    str60_set(task, "WARNING: ROUNDING ERRORS PREVENT PROGRESS");
  if (brackt && stmax - stmin <= xtol * stmax) str60_set(task, "WARNING: XTOL TEST SATISFIED");
  if (stp == stpmax && f <= ftest && g <= gtest) str60_set(task, "WARNING: STP = STPMAX");
  if (stp == stpmin && (f > ftest || g >= gtest)) str60_set(task, "WARNING: STP = STPMIN");
  if (f <= ftest && std::fabs(g) <= gtol * (-ginit)) str60_set(task, "CONVERGENCE");
  if (str60_pre(task, "WARN") || str60_pre(task, "CONV")) {
    save();
    return;
  }

  if (stage == 1 && f <= fx && f > ftest) {
    double fm = f - stp * gtest, fxm = fx - stx * gtest, fym = fy - sty * gtest;
    double gm = g - gtest, gxm = gx - gtest, gym = gy - gtest;
    dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax);
    fx = fxm + stx * gtest;
    fy = fym + sty * gtest;
    gx = gxm + gtest;
    gy = gym + gtest;
  } else {
    dcstep(stx, fx, gx, sty, fy, gy, stp, f, g, brackt, stmin, stmax);
  }
  if (brackt) {
    if (std::fabs(sty - stx) >= p66 * width1) stp = stx + p5 * (sty - stx);
    width1 = width;
    width = std::fabs(sty - stx);
  }
  if (brackt) {
    stmin = std::min(stx, sty);
    stmax = std::max(stx, sty);
  } else {
    stmin = stp + xtrapl * (stp - stx);
    stmax = stp + xtrapu * (stp - stx);
  }
  stp = std::max(stp, stpmin);
  stp = std::min(stp, stpmax);
  if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax))
    stp = stx;
  str60_set(task, "FG");
  save();
}

// hpsolb (src/lbfgsb.f90:2079-2157): t(1:n) / iorder(1:n) hold the breakpoints not yet used.
// iheap == 0: first rearrange them into a min-heap by sifting t(2), t(3), ... up; then move the
// least element to t(n) and restore the heap on t(1:n-1).  Used by the replay of the reference's
// pop order among EQUAL breakpoints (a walk that ends inside a tie group, solver.hip exact_init);
// arrays are 0-based here, i and j keep the reference's 1-based meaning.  I = the type of the row
// numbers (uint32_t while n_global < 2^32, int64_t beyond).
template <typename I>
inline void hpsolb(int64_t n, double *t, I *iorder, int iheap) {
  if (iheap == 0) {
    for (int64_t k = 2; k <= n; ++k) {
      const double ddum = t[k - 1];
      const I indxin = iorder[k - 1];
      int64_t i = k;
      while (i > 1) {
        const int64_t j = i / 2;
        if (!(ddum < t[j - 1])) break;
        t[i - 1] = t[j - 1];
        iorder[i - 1] = iorder[j - 1];
        i = j;
      }
      t[i - 1] = ddum;
      iorder[i - 1] = indxin;
    }
  }
  if (n > 1) {
    int64_t i = 1;
    const double out = t[0];
    const I indxou = iorder[0];
    const double ddum = t[n - 1];
    const I indxin = iorder[n - 1];
    for (;;) {
      int64_t j = i + i;
      if (j > n - 1) break;
      if (t[j] < t[j - 1]) j = j + 1;  // t(j+1) < t(j)
      if (!(t[j - 1] < ddum)) break;
      t[i - 1] = t[j - 1];
      iorder[i - 1] = iorder[j - 1];
      i = j;
    }
    t[i - 1] = ddum;
    iorder[i - 1] = indxin;
    t[n - 1] = out;
    iorder[n - 1] = indxou;
  }
}

}  // namespace lbh
