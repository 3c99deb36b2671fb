// report.hpp -- the reference's text output (prn1lb, prn2lb, prn3lb,
// reference src/lbfgsb.f90:2363-2579) restated on the host so that
// test/driver1.f90's transcript and its iteration file can be diffed against
// test/OUTPUTS/output_90_1 and iterate.dat.  Host-only text I/O; no n-work
// except optional D2H dumps of x and g when iprint > 100.
//
// Fortran edit descriptors are emulated: 1p,dW.D -> "d.dddD+ee", 1p,eW.D ->
// "d.dddE+ee", iW -> right-justified integers; list-directed output follows the
// gfortran layout of the golden transcripts (12-wide integers).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace lbr {

// 1p scale factor with D or E exponent letter, width w, d fraction digits
inline std::string fexp(double v, int w, int d, char letter) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.*E", d, v);
  std::string s(buf);
  // C gives at least 2 exponent digits, like Fortran for |exp| < 100
  for (auto &c : s)
    if (c == 'E') c = letter;
  if (std::isnan(v)) s = "NaN";
  if ((int)s.size() < w) s = std::string(w - s.size(), ' ') + s;
  return s;
}
inline std::string fD(double v, int w, int d) { return fexp(v, w, d, 'D'); }
inline std::string fE(double v, int w, int d) { return fexp(v, w, d, 'E'); }

// list-directed real(8) in gfortran style: 1.083490083461424E-009 -> we print 17
// significant digits with a 3-digit exponent
inline std::string flist(double v) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.15E", v);
  std::string s(buf);
  size_t e = s.find('E');
  if (e != std::string::npos) {
    std::string mant = s.substr(0, e);
    int ex = std::atoi(s.c_str() + e + 1);
    char eb[16];
    std::snprintf(eb, sizeof eb, "E%c%03d", ex < 0 ? '-' : '+', std::abs(ex));
    s = mant + eb;
  }
  return "   " + s;
}

struct Report {
  FILE *out = stdout;
  FILE *itf = nullptr;

  // n-vector dumps of iprint >= 100 (:2404-2408, :2449-2452, :2511-2514):
  // '(/,a4,1p,6(1x,d11.4),/,(4x,1p,6(1x,d11.4)))'
  void vec_a4(const char *label, const double *v, long long n) {
    std::fprintf(out, "\n%4s", label);
    for (long long i = 0; i < n; ++i) {
      if (i > 0 && i % 6 == 0) std::fprintf(out, "\n    ");
      std::fprintf(out, " %s", fD(v[i], 11, 4).c_str());
    }
    std::fprintf(out, "\n");
  }
  // '(A,/,(4x,1p,6(1x,d11.4)))' (:1345, :1527)
  void vec_rows(const char *label, const double *v, long long n) {
    std::fprintf(out, "%s", label);
    for (long long i = 0; i < n; ++i) {
      if (i % 6 == 0) std::fprintf(out, "\n    ");
      std::fprintf(out, " %s", fD(v[i], 11, 4).c_str());
    }
    std::fprintf(out, "\n");
  }
  // cauchy's per-segment report (:1408-1412, :1502-1508)
  void piece(int nseg, double f1, double f2) {
    std::fprintf(out, "Piece    %3d --f1, f2 at start point  %s %s\n", nseg, fD(f1, 11, 4).c_str(),
                 fD(f2, 11, 4).c_str());
  }

  // prn1lb :2363-2412
  void prn1lb(long long n, int m, int iprint, double epsmch) {
    if (iprint < 0) return;
    std::fprintf(out, "RUNNING THE L-BFGS-B CODE\n\n           * * *\n\nMachine precision =%s\n",
                 fD(epsmch, 10, 3).c_str());
    std::fprintf(out, " N = %12lld     M = %12d\n", n, m);
    if (iprint >= 1 && itf) {
      std::fprintf(itf,
                   "RUNNING THE L-BFGS-B CODE\n\n"
                   "it    = iteration number\n"
                   "nf    = number of function evaluations\n"
                   "nseg  = number of segments explored during the Cauchy search\n"
                   "nact  = number of active bounds at the generalized Cauchy point\n"
                   "sub   = manner in which the subspace minimization terminated:\n"
                   "        con = converged, bnd = a bound was reached\n"
                   "itls  = number of iterations performed in the line search\n"
                   "stepl = step length used\n"
                   "tstep = norm of the displacement (total step)\n"
                   "projg = norm of the projected gradient\n"
                   "f     = function value\n\n"
                   "           * * *\n\n"
                   "Machine precision =%s\n",
                   fD(epsmch, 10, 3).c_str());
      std::fprintf(itf, " N = %12lld     M = %12d\n", n, m);
      std::fprintf(itf, "\n   it   nf  nseg  nact  sub  itls  stepl    tstep     projg        f\n");
    }
  }

  void active_msgs(int iprint, bool prjctd, bool cnstnd, long long nbdd) {  // :1031-1038
    if (iprint >= 0) {
      if (prjctd)
        std::fprintf(out, " The initial X is infeasible.  Restart with its projection.\n");
      if (!cnstnd) std::fprintf(out, " This problem is unconstrained.\n");
    }
    if (iprint > 0)
      std::fprintf(out, "\nAt X0 %9lld variables are exactly at the bounds\n", nbdd);
  }

  void iterate0(int iprint, int iter, int nfgv, double f, double sbgnrm) {  // :584-589
    if (iprint < 1) return;
    std::fprintf(out, "\nAt iterate%5d    f= %s    |proj g|= %s\n", iter, fD(f, 12, 5).c_str(),
                 fD(sbgnrm, 12, 5).c_str());
    if (itf)
      std::fprintf(itf, " %4d %4d     -     -   -     -     -        -    %s %s\n", iter, nfgv,
                   fD(sbgnrm, 10, 3).c_str(), fD(f, 10, 3).c_str());
  }

  // prn2lb :2428-2461 (vector dumps for iprint>100 are done by the caller)
  void prn2lb(int iprint, int iter, int nfgv, int nact, double sbgnrm, int nseg,
              const char *word, int iback, double stp, double xstep, double f) {
    if (iprint >= 99) {
      std::fprintf(out, " LINE SEARCH %11d  times; norm of step = %s\n", iback,
                   flist(xstep).c_str());
      std::fprintf(out, "\nAt iterate%5d    f= %s    |proj g|= %s\n", iter,
                   fD(f, 12, 5).c_str(), fD(sbgnrm, 12, 5).c_str());
    } else if (iprint > 0) {
      if (iter % iprint == 0)
        std::fprintf(out, "\nAt iterate%5d    f= %s    |proj g|= %s\n", iter,
                     fD(f, 12, 5).c_str(), fD(sbgnrm, 12, 5).c_str());
    }
    if (iprint >= 1 && itf)
      std::fprintf(itf, " %4d %4d %5d %5d  %3s %4d  %s  %s %s %s\n", iter, nfgv, nseg, nact, word,
                   iback, fD(stp, 7, 1).c_str(), fD(xstep, 7, 1).c_str(),
                   fD(sbgnrm, 10, 3).c_str(), fD(f, 10, 3).c_str());
  }

  static const char *info_text(int info) {
    switch (info) {
      case -1: return "\n Matrix in 1st Cholesky factorization in formk is not Pos. Def.\n";
      case -2: return "\n Matrix in 2st Cholesky factorization in formk is not Pos. Def.\n";
      case -3: return "\n Matrix in the Cholesky factorization in formt is not Pos. Def.\n";
      case -4:
        return "\n Derivative >= 0, backtracking line search impossible.\n"
               "   Previous x, f and g restored.\n"
               " Possible causes: 1 error in function or gradient evaluation;\n"
               "                  2 rounding errors dominate computation.\n";
      case -5:
        return "\n Warning:  more than 10 function and gradient\n"
               "   evaluations in the last line search.  Termination\n"
               "   may possibly be caused by a bad search direction.\n";
      case -8: return "\n The triangular system is singular.\n";
      case -9:
        return "\n Line search cannot locate an adequate point after 20 function\n"
               "  and gradient evaluations.  Previous x, f and g restored.\n"
               " Possible causes: 1 error in function or gradient evaluation;\n"
               "                  2 rounding error dominate computation.\n";
    }
    return nullptr;
  }

  // prn3lb :2478-2579
  void prn3lb(long long n, double f, const char *task60, int iprint, int info, int iter, int nfgv,
              int nintol, int nskip, int nact, double sbgnrm, double time, int nseg,
              const char *word, int iback, double stp, double xstep, long long k, double cachyt,
              double sbtime, double lnscht, const double *xfinal = nullptr) {
    const bool err = std::strncmp(task60, "ERROR", 5) == 0;
    if (!err && iprint >= 0) {
      std::fprintf(out,
                   "\n           * * *\n\n"
                   "Tit   = total number of iterations\n"
                   "Tnf   = total number of function evaluations\n"
                   "Tnint = total number of segments explored during Cauchy searches\n"
                   "Skip  = number of BFGS updates skipped\n"
                   "Nact  = number of active bounds at final generalized Cauchy point\n"
                   "Projg = norm of the final projected gradient\n"
                   "F     = final function value\n\n"
                   "           * * *\n");
      std::fprintf(out, "\n   N    Tit     Tnf  Tnint  Skip  Nact     Projg        F\n");
      std::fprintf(out, "%5lld %6d %6d %6d  %4d %5d  %s  %s\n", n, iter, nfgv, nintol, nskip, nact,
                   fD(sbgnrm, 10, 3).c_str(), fD(f, 10, 3).c_str());
      if (iprint >= 100 && xfinal) vec_a4("X =", xfinal, n);  // :2511-2514
      if (iprint >= 1) std::fprintf(out, "  F =%s\n", flist(f).c_str());
    }
    if (iprint >= 0) {
      std::fprintf(out, "\n%.60s\n", task60);
      if (info == -6)
        std::fprintf(out, "  Input nbd(%12lld ) is invalid.\n", k);
      else if (info == -7)
        std::fprintf(out, "  l(%12lld ) > u(%12lld ).  No feasible solution.\n", k, k);
      else if (const char *t = info_text(info))
        std::fputs(t, out);
      if (iprint >= 1)
        std::fprintf(out,
                     "\n Cauchy                time%s seconds.\n Subspace minimization time%s "
                     "seconds.\n Line search           time%s seconds.\n",
                     fE(cachyt, 10, 3).c_str(), fE(sbtime, 10, 3).c_str(),
                     fE(lnscht, 10, 3).c_str());
      std::fprintf(out, "\n Total User time%s seconds.\n\n", fE(time, 10, 3).c_str());
      if (iprint >= 1 && itf) {
        if (info == -4 || info == -9)
          std::fprintf(itf, " %4d %4d %5d %5d  %3s %4d  %s  %s      -          -\n", iter, nfgv,
                       nseg, nact, word, iback, fD(stp, 7, 1).c_str(), fD(xstep, 7, 1).c_str());
        std::fprintf(itf, "\n%.60s\n", task60);
        if (info != -4 && info != -6 && info != -7)
          if (const char *t = info_text(info)) std::fputs(t, itf);
        if (info == -4) std::fputs(info_text(-4), out);  // reference :2559 writes to stdout
        std::fprintf(itf, "\n Total User time%s seconds.\n\n", fE(time, 10, 3).c_str());
        std::fflush(itf);
      }
      std::fflush(out);
    }
  }
};

}  // namespace lbr
