/* Plain-C client of the drop-in boundary (include/gpmi.h): fits a small Gaussian process and predicts, with
 * nothing but the C-ABI - no C++, no Python, no torch.  Built by tests/test_c_abi.py with gcc:
 *   gcc -std=c99 -Wall -Iinclude tests/c_abi/fit_predict.c -o fit_predict -Linference-tools_amd/inference_amd/lib -lgpmi -lm
 * Exit code 0 and a line "ok ..." when the self-check passes (K alpha = y - mu to 1e-10 with K rebuilt here). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpmi.h"

#define N 300
#define D 2
#define M 5

static double unit(unsigned* s) { /* small LCG: the data only has to be reproducible */
  *s = *s * 1664525u + 1013904223u;
  return (double)(*s >> 8) / 16777216.0;
}

int main(void) {
  static double x[N * D], y[N], noise[N], mu[N], alpha[N], pts[M * D], pm[M], pv[M];
  unsigned seed = 12345u;
  int i, j, k, info = -1, ndev = 0;
  double theta[1 + D] = {0.0, log(0.4), log(0.6)}; /* ln a, ln l_1, ln l_2 */
  double logdet = 0.0, lml = 0.0, worst = 0.0, scale = 0.0;
  gpmi_ctx* ctx = NULL;

  if (gpmi_device_count(&ndev) != GPMI_OK || ndev < 1) {
    printf("skip: no device\n");
    return 77;
  }
  for (i = 0; i < N; ++i) {
    for (k = 0; k < D; ++k) x[i * D + k] = unit(&seed);
    y[i] = sin(4.0 * x[i * D]) * cos(3.0 * x[i * D + 1]) + 0.05 * (unit(&seed) - 0.5);
    noise[i] = 0.05 * 0.05;
    mu[i] = 0.1;
  }
  for (i = 0; i < M * D; ++i) pts[i] = unit(&seed);

  if (gpmi_create(0, &ctx) != GPMI_OK) {
    fprintf(stderr, "gpmi_create: %s\n", gpmi_last_error(NULL));
    return 1;
  }
  if (gpmi_set_data(ctx, x, y, noise, NULL, N, D) != GPMI_OK ||
      gpmi_fit(ctx, GPMI_KERNEL_SE, theta, 1 + D, 0.0, mu, alpha, &logdet, &info) != GPMI_OK || info != 0 ||
      gpmi_lml(ctx, GPMI_KERNEL_SE, theta, 1 + D, 0.0, mu, &lml, &info) != GPMI_OK || info != 0 ||
      gpmi_predict(ctx, pts, M, pm, pv) != GPMI_OK) {
    fprintf(stderr, "gpmi call failed (info %d): %s\n", info, gpmi_last_error(ctx));
    return 1;
  }
  /* self-check: (K + noise) alpha = y - mu with K = a^2 (exp(-1/2 sum ((x_i - x_j) / l)^2) + 1e-12 delta_ij) */
  for (i = 0; i < N; ++i) {
    double s = 0.0;
    for (j = 0; j < N; ++j) {
      double z = 0.0, kij;
      for (k = 0; k < D; ++k) {
        const double dx = (x[i * D + k] - x[j * D + k]) / exp(theta[1 + k]);
        z += 0.5 * dx * dx;
      }
      kij = exp(2.0 * theta[0]) * (exp(-z) + (i == j ? 1e-12 : 0.0)) + (i == j ? noise[i] : 0.0);
      s += kij * alpha[j];
    }
    if (fabs(s - (y[i] - mu[i])) > worst) worst = fabs(s - (y[i] - mu[i]));
    if (fabs(y[i] - mu[i]) > scale) scale = fabs(y[i] - mu[i]);
  }
  gpmi_destroy(ctx);
  if (!(worst <= 1e-10 * scale) || !(pv[0] >= 0.0) || !isfinite(lml)) {
    fprintf(stderr, "self-check failed: residual %.3e (scale %.3e), var %.3e, lml %.6f\n", worst, scale, pv[0], lml);
    return 1;
  }
  printf("ok residual %.2e lml %.6f logdet %.6f mu*[0] %.6f var*[0] %.6e\n", worst / scale, lml, logdet, pm[0] + 0.1, pv[0]);
  return 0;
}
