// Test-only C doorway into the PRODUCT's host-side 2m x 2m algebra (lbfgsb_amd/csrc/
// host_dense.hpp) and its Fortran-format printing (report.hpp), so that the CPU test-suite can
// compare them with the oracle without a GPU.  Built by tests/test_host_logic_cpu.py with g++.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../lbfgsb_amd/csrc/host_dense.hpp"
#include "../lbfgsb_amd/csrc/report.hpp"

extern "C" {
int hd_dpofa(double *a, int lda, int n) { return lbh::dpofa(lbh::Mat{a, lda}, n); }
int hd_dtrsl(double *t, int ldt, int n, double *b, int job) { return lbh::dtrsl(lbh::Mat{t, ldt}, n, b, job); }
int hd_bmv(int m, const double *sy, const double *wt, int col, const double *v, double *p) {
  return lbh::bmv(m, sy, wt, col, v, p);
}
int hd_formt(int m, double *wt, const double *sy, const double *ss, int col, double theta) {
  return lbh::formt(m, wt, sy, ss, col, theta);
}
void hd_dcsrch(double f, double g, double *stp, double ftol, double gtol, double xtol, double stpmin,
               double stpmax, char *task, int32_t *isave, double *dsave) {
  lbh::dcsrch(f, g, *stp, ftol, gtol, xtol, stpmin, stpmax, task, isave, dsave);
}
void hd_hpsolb32(int64_t n, double *t, uint32_t *iorder, int iheap) { lbh::hpsolb(n, t, iorder, iheap); }
void hd_hpsolb64(int64_t n, double *t, int64_t *iorder, int iheap) { lbh::hpsolb(n, t, iorder, iheap); }
void hd_fmt(double v, int w, int d, int letter, char *out, int cap) {
  std::string s = lbr::fexp(v, w, d, (char)letter);
  std::snprintf(out, cap, "%s", s.c_str());
}
}
