/* examples/driver1.c -- the reference's test/driver1.f90 as a plain C program over the C ABI
 * (include/lbfgsb_hip.h): extended Rosenbrock, n = 25, m = 5, the bounds and the start of
 * test/driver1.f90:233-251, the objective of :274-289, the reverse-communication loop of :263-292.
 * The host-pointer entry takes the reference's own argument list, so the work arrays are sized
 * exactly as the reference documents them (src/lbfgsb.f90:92-186); the iteration runs on the GPU.
 *
 *   cc -std=c99 -Iinclude examples/driver1.c -Llbfgsb_amd -llbfgsb_hip -Wl,-rpath,$PWD/lbfgsb_amd -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lbfgsb_hip.h"

int main(void) {
  enum { N = 25, M = 5 };
  double x[N], l[N], u[N], g[N], f = 0.0;
  int32_t nbd[N];
  double *wa = (double *)calloc(2 * M * N + 5 * N + 11 * M * M + 8 * M, sizeof(double));
  int32_t *iwa = (int32_t *)calloc(3 * N, sizeof(int32_t));
  char task[60], csave[60];
  int32_t lsave[4] = {0, 0, 0, 0}, isave[44];
  double dsave[29];
  const double factr = 1.0e7, pgtol = 1.0e-5;
  int i, rc;
  memset(isave, 0, sizeof isave);
  memset(dsave, 0, sizeof dsave);
  for (i = 0; i < N; i += 2) {          /* odd-numbered variables (1-based) */
    nbd[i] = 2, l[i] = 1.0, u[i] = 100.0;
  }
  for (i = 1; i < N; i += 2) {          /* even-numbered variables */
    nbd[i] = 2, l[i] = -100.0, u[i] = 100.0;
  }
  for (i = 0; i < N; ++i) x[i] = 3.0;
  memset(task, ' ', 60);
  memset(csave, ' ', 60);
  memcpy(task, "START", 5);
  while (!strncmp(task, "FG", 2) || !strncmp(task, "NEW_X", 5) || !strncmp(task, "START", 5)) {
    rc = lbfgsb_hip_setulb_host(N, M, x, l, u, nbd, &f, g, factr, pgtol, wa, iwa, task, -1, csave, lsave,
                                isave, dsave, NULL, 8, 0);
    if (rc != LBFGSB_OK) {
      fprintf(stderr, "lbfgsb_hip_setulb_host: %d (%s)\n", rc, lbfgsb_hip_last_error());
      return 1;
    }
    if (!strncmp(task, "FG", 2)) {
      double t1, t2;
      f = 0.25 * (x[0] - 1.0) * (x[0] - 1.0);
      for (i = 1; i < N; ++i) f += (x[i] - x[i - 1] * x[i - 1]) * (x[i] - x[i - 1] * x[i - 1]);
      f *= 4.0;
      t1 = x[1] - x[0] * x[0];
      g[0] = 2.0 * (x[0] - 1.0) - 16.0 * x[0] * t1;
      for (i = 1; i < N - 1; ++i) {
        t2 = t1;
        t1 = x[i + 1] - x[i] * x[i];
        g[i] = 8.0 * t2 - 16.0 * x[i] * t1;
      }
      g[N - 1] = 8.0 * t1;
    }
  }
  printf("task = %.48s\n", task);
  printf("iterations = %d  nfg = %d  f = %.16e  |proj g| = %.6e\n", (int)isave[29], (int)isave[33], f, dsave[12]);
  free(wa);
  free(iwa);
  return 0;
}
