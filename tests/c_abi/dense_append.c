/* Plain-C client of the dense (plugin-kernel) entry points and of the O(n^2) append of a training point
 * (include/gpmi.h: gpmi_fit_dense, gpmi_lml_dense, gpmi_predict_dense, gpmi_set_option, gpmi_append_point,
 * gpmi_capacity).  Built by tests/test_c_abi.py with gcc -std=c99 -pedantic -Werror.
 * Part 1: a Matern-3/2 covariance the library has no device code for is built HERE and handed over dense; checks:
 *   K alpha = y - mu,  K^-1 K = 1 (row sums),  |L^-1 k_q|^2 = k_q . K^-1 k_q,  k_q . alpha against a host dot product.
 * Part 2: a SquaredExponential fit with room for more points, one point appended: (K' + noise') alpha' = y' - mu'
 *   on the enlarged data, with K' rebuilt here.
 * Exit code 0 and a line "ok ..." when every check holds to 1e-9 (relative). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpmi.h"

#define N 200
#define D 2
#define M 4

static double unit(unsigned* s) {
  *s = *s * 1664525u + 1013904223u;
  return (double)(*s >> 8) / 16777216.0;
}

static double matern32(const double* a, const double* b) {
  double r2 = 0.0, r;
  int k;
  for (k = 0; k < D; ++k) r2 += (a[k] - b[k]) * (a[k] - b[k]) / (0.5 * 0.5);
  r = sqrt(3.0 * r2);
  return 1.3 * (1.0 + r) * exp(-r);
}

static double se(const double* a, const double* b, const double* theta) {
  double z = 0.0;
  int k;
  for (k = 0; k < D; ++k) {
    const double dx = (a[k] - b[k]) / exp(theta[1 + k]);
    z += 0.5 * dx * dx;
  }
  return exp(2.0 * theta[0]) * exp(-z);
}

int main(void) {
  static double x[(N + 1) * D], y[N + 1], noise[N + 1], mu[N + 1], alpha[N + 1];
  static double K[N * N], iK[N * N], Kq[M * N], pts[M * D], kal[M], ssq[M];
  unsigned seed = 4242u;
  int i, j, q, info = -1, ndev = 0;
  double theta[1 + D] = {0.1, -0.9, -0.6};
  double logdet = 0.0, lml = 0.0, worst = 0.0, scale = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0, w5 = 0.0, s5 = 0.0;
  int64_t cap = 0;
  gpmi_ctx* ctx = NULL;

  if (gpmi_device_count(&ndev) != GPMI_OK || ndev < 1) {
    printf("skip: no device\n");
    return 77;
  }
  for (i = 0; i < N + 1; ++i) {
    x[i * D] = unit(&seed);
    x[i * D + 1] = unit(&seed);
    y[i] = sin(5.0 * x[i * D]) + cos(4.0 * x[i * D + 1]) + 0.05 * (unit(&seed) - 0.5);
    noise[i] = 0.04 * 0.04;
    mu[i] = 0.2;
  }
  for (i = 0; i < M * D; ++i) pts[i] = unit(&seed);

  /* ---- part 1: dense entry points --------------------------------------------------------------------- */
  for (i = 0; i < N; ++i)
    for (j = 0; j < N; ++j) K[i * N + j] = matern32(x + i * D, x + j * D) + (i == j ? noise[i] : 0.0);
  for (q = 0; q < M; ++q)
    for (j = 0; j < N; ++j) Kq[q * N + j] = matern32(pts + q * D, x + j * D);
  if (gpmi_create(0, &ctx) != GPMI_OK) {
    fprintf(stderr, "gpmi_create: %s\n", gpmi_last_error(NULL));
    return 1;
  }
  if (gpmi_set_data(ctx, x, y, noise, NULL, N, D) != GPMI_OK ||
      gpmi_fit_dense(ctx, K, mu, alpha, &logdet, &info) != GPMI_OK || info != 0 ||
      gpmi_predict_dense(ctx, Kq, M, kal, ssq) != GPMI_OK ||
      gpmi_lml_dense(ctx, K, mu, &lml, NULL, iK, &info) != GPMI_OK || info != 0) {
    fprintf(stderr, "dense call failed (info %d): %s\n", info, gpmi_last_error(ctx));
    return 1;
  }
  for (i = 0; i < N; ++i) {
    double s = 0.0, rowsum = 0.0;
    for (j = 0; j < N; ++j) {
      double e = 0.0;
      int k;
      s += K[i * N + j] * alpha[j];
      for (k = 0; k < N; ++k) e += iK[i * N + k] * K[k * N + j]; /* (K^-1 K)_ij */
      rowsum += fabs(e - (i == j ? 1.0 : 0.0));
    }
    if (fabs(s - (y[i] - mu[i])) > worst) worst = fabs(s - (y[i] - mu[i]));
    if (fabs(y[i] - mu[i]) > scale) scale = fabs(y[i] - mu[i]);
    if (rowsum > w2) w2 = rowsum;
  }
  for (q = 0; q < M; ++q) {
    double dot = 0.0, quad = 0.0;
    for (i = 0; i < N; ++i) {
      double t = 0.0;
      dot += Kq[q * N + i] * alpha[i];
      for (j = 0; j < N; ++j) t += iK[i * N + j] * Kq[q * N + j];
      quad += Kq[q * N + i] * t;
    }
    if (fabs(dot - kal[q]) > w3 * 1.0) w3 = fabs(dot - kal[q]) / (fabs(dot) + 1e-300);
    if (fabs(quad - ssq[q]) / quad > w4) w4 = fabs(quad - ssq[q]) / quad;
  }
  gpmi_destroy(ctx);
  ctx = NULL;

  /* ---- part 2: append one training point at fixed hyper-parameters ------------------------------------ */
  if (gpmi_create(0, &ctx) != GPMI_OK || gpmi_set_option(ctx, GPMI_OPT_RESERVE_POINTS, 8) != GPMI_OK ||
      gpmi_set_data(ctx, x, y, noise, NULL, N, D) != GPMI_OK || gpmi_capacity(ctx, &cap) != GPMI_OK || cap < N + 1 ||
      gpmi_fit(ctx, GPMI_KERNEL_SE, theta, 1 + D, 0.0, mu, alpha, &logdet, &info) != GPMI_OK || info != 0 ||
      gpmi_append_point(ctx, x + N * D, y[N], noise[N], mu, alpha, &logdet, &info) != GPMI_OK || info != 0) {
    fprintf(stderr, "append failed (info %d, capacity %ld): %s\n", info, (long)cap, gpmi_last_error(ctx));
    return 1;
  }
  for (i = 0; i < N + 1; ++i) {
    double s = 0.0;
    for (j = 0; j < N + 1; ++j)
      s += (se(x + i * D, x + j * D, theta) + (i == j ? exp(2.0 * theta[0]) * 1e-12 + noise[i] : 0.0)) * alpha[j];
    if (fabs(s - (y[i] - mu[i])) > w5) w5 = fabs(s - (y[i] - mu[i]));
    if (fabs(y[i] - mu[i]) > s5) s5 = fabs(y[i] - mu[i]);
  }
  gpmi_destroy(ctx);
  if (!(worst <= 1e-9 * scale) || !(w2 <= 1e-7) || !(w3 <= 1e-9) || !(w4 <= 1e-8) || !(w5 <= 1e-9 * s5) || !isfinite(lml)) {
    fprintf(stderr, "self-check failed: K alpha %.2e, K^-1 K %.2e, kq.alpha %.2e, |L^-1 kq|^2 %.2e, append %.2e, lml %.6f\n",
            worst / scale, w2, w3, w4, w5 / s5, lml);
    return 1;
  }
  printf("ok dense: K alpha %.1e, K^-1 K %.1e, kq.alpha %.1e, quad %.1e | append: %.1e | lml %.6f capacity %ld\n",
         worst / scale, w2, w3, w4, w5 / s5, lml, (long)cap);
  return 0;
}
