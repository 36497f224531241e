/* Plain C99 against include/rtd.h -- no Python, no C++: the drop-in boundary driven the way a binding in any language would
 * (the reference's counterpart is one pydisort() call per column, src/PythonicDISORT/pydisort.py:13-29, followed by calls of the
 * returned closures u / flux_up / flux_down, _assemble_intensity_and_fluxes.py:170-613).
 *
 *   ./solve_columns [ncols]      three-layer Henyey-Greenstein atmospheres, 16 streams, delta-M scaling, beam source;
 *                                prints flux_up, flux_down (diffuse, direct) at tau = 0 and u at one depth per column.
 *
 * What the host does: Gauss-Legendre nodes on [0, 1] (the reference's double-Gauss quadrature, pydisort.py:304) and the raw inputs.
 * What the device does: everything else (rtd_plan_set_columns_raw: delta-M scaling and source rescaling; rtd_plan_solve: eigen stage
 * and boundary-condition solve; rtd_plan_evaluate: the closures).
 * Build: gcc -std=c99 -O2 -I include examples/solve_columns.c pythonic-disort_amd/pydisort_amd/librtd.so -lm -o solve_columns
 * tests/test_gpu_c_example.py compiles it, runs it and compares its output with the Python front end. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rtd.h"

/* nodes and weights of the n-point Gauss-Legendre rule mapped to [0, 1] (Newton iteration on P_n) */
static void gauss_legendre_01(int n, double* x, double* w) {
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < n; ++i) {
    double z = cos(pi * (i + 0.75) / (n + 0.5)), pp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p0 = 1.0, p1 = z;
      for (int k = 2; k <= n; ++k) {
        const double p2 = ((2.0 * k - 1.0) * z * p1 - (k - 1.0) * p0) / k;
        p0 = p1;
        p1 = p2;
      }
      pp = n * (z * p1 - p0) / (z * z - 1.0);
      const double dz = p1 / pp;
      z -= dz;
      if (fabs(dz) < 1e-16) break;
    }
    x[n - 1 - i] = 0.5 * (z + 1.0); /* ascending, as numpy's leggauss */
    w[n - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
  }
}

#define CHECK(call)                                                                  \
  do {                                                                               \
    const int rc_ = (call);                                                          \
    if (rc_ != RTD_OK) {                                                             \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rtd_last_error());         \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 3, L = 3, NQ = 16, N = NQ / 2, NLEG_ALL = NQ + 1;
  int32_t ndev = 0;
  CHECK(rtd_device_count(&ndev));
  if (ndev < 1) {
    fprintf(stderr, "no HIP device\n");
    return 2;
  }
  double mu[8], wt[8];
  gauss_legendre_01(N, mu, wt);
  double* tau = malloc(sizeof(double) * C * L);
  double* omega = malloc(sizeof(double) * C * L);
  double* f = malloc(sizeof(double) * C * L);
  double* leg = malloc(sizeof(double) * C * L * NLEG_ALL);
  double *mu0 = malloc(sizeof(double) * C), *I0 = malloc(sizeof(double) * C), *phi0 = malloc(sizeof(double) * C);
  for (int c = 0; c < C; ++c) {
    mu0[c] = 0.3 + 0.6 * (c + 1.0) / (C + 1.0);
    I0[c] = 3.0;
    phi0[c] = 0.5;
    for (int l = 0; l < L; ++l) {
      const double g = 0.55 + 0.1 * l + 0.01 * c;
      tau[c * L + l] = 0.4 * (l + 1) * (1.0 + 0.05 * c);       /* cumulative optical depth of the layer bottoms */
      omega[c * L + l] = 0.95 - 0.1 * l;
      for (int k = 0; k < NLEG_ALL; ++k) leg[(c * L + l) * NLEG_ALL + k] = pow(g, k); /* Henyey-Greenstein moments */
      f[c * L + l] = pow(g, NQ);                                                        /* delta-M truncation fraction */
    }
  }
  rtd_dims dims = {C, L, NQ, NQ, NQ, 0, 0, 1};
  rtd_plan* plan = NULL;
  CHECK(rtd_plan_create(&dims, 0, &plan));
  CHECK(rtd_plan_set_quadrature(plan, mu, wt));
  CHECK(rtd_plan_set_columns_raw(plan, tau, omega, leg, NLEG_ALL, f, mu0, I0, phi0, NULL, NULL, NULL, NULL, NULL));
  CHECK(rtd_plan_solve(plan));
  const int ntau = 2, nphi = 2;
  double* pts = malloc(sizeof(double) * C * ntau);
  const double phi[2] = {0.0, 2.0};
  for (int c = 0; c < C; ++c) {
    pts[c * ntau] = 0.0;
    pts[c * ntau + 1] = 0.37 * tau[c * L + L - 1];
  }
  double* u = malloc(sizeof(double) * C * NQ * ntau * nphi);
  double *fup = malloc(sizeof(double) * C * ntau), *fdn = malloc(sizeof(double) * C * ntau), *fdir = malloc(sizeof(double) * C * ntau);
  CHECK(rtd_plan_evaluate(plan, ntau, pts, nphi, phi, 0, u, NULL, fup, fdn, fdir, NULL));
  for (int c = 0; c < C; ++c)
    printf("column %d: flux_up(0) %.17g flux_down(0) %.17g + %.17g  u[mu_0, tau_1, phi_1] %.17g  u[-mu_0, tau_1, phi_0] %.17g\n", c,
           fup[c * ntau], fdn[c * ntau], fdir[c * ntau], u[((c * NQ + 0) * ntau + 1) * nphi + 1], u[((c * NQ + N) * ntau + 1) * nphi + 0]);
  CHECK(rtd_plan_destroy(plan));
  free(tau); free(omega); free(f); free(leg); free(mu0); free(I0); free(phi0); free(pts); free(u); free(fup); free(fdn); free(fdir);
  return 0;
}
