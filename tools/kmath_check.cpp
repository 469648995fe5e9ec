// Host-side accuracy check of csrc/kmath.h against long double:
//   g++ -O2 -std=c++17 -ffp-contract=off -I inference-tools_amd/csrc tools/kmath_check.cpp -o /tmp/kmath_check && /tmp/kmath_check
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "kmath.h"

static double ulp_err(double got, long double want) {
  if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
  double w = (double)want;
  double u = std::nextafter(std::fabs(w), INFINITY) - std::fabs(w);
  if (u == 0.0) u = 4.9e-324;
  return (double)(std::fabs((long double)got - want) / u);
}

int main() {
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const int T = 10000000;
  double worst_e = 0, worst_l = 0, worst_p = 0, worst_p_rel = 0, worst_p_scaled = 0;
  for (int t = 0; t < T; t += 4) {
    double x[4], o[4];
    for (int i = 0; i < 4; ++i) {
      const double u = U(g);
      x[i] = (t % 3 == 0) ? -745.0 * u : (t % 3 == 1 ? -40.0 * u : -std::exp(-30.0 * u));
    }
    kmath::exp_neg(x, o);
    for (int i = 0; i < 4; ++i) {
      const long double w = expl((long double)x[i]);
      if (w > 2.3e-308L) worst_e = std::fmax(worst_e, ulp_err(o[i], w));
    }
    double z[4], l[4];
    for (int i = 0; i < 4; ++i) {
      const double u = U(g);
      const int m = (t / 4) % 4;
      z[i] = (m == 0) ? std::exp(690.0 * u) : (m == 1 ? 10.0 * u : (m == 2 ? std::exp(-40.0 * u) : u * 3.0));
    }
    kmath::log1p_pos(z, l);
    for (int i = 0; i < 4; ++i) worst_l = std::fmax(worst_l, ulp_err(l[i], log1pl((long double)z[i])));
    // the RationalQuadratic power: (1 + s / kappa)^-kappa, kappa in [e^-2, e^6], s in [0, 200]
    double zz[4], ll[4], ee[4], pp[4], kap[4], ss[4];
    for (int i = 0; i < 4; ++i) {
      kap[i] = std::exp(-2.0 + 8.0 * U(g));
      ss[i] = 200.0 * U(g) * U(g);
      zz[i] = ss[i] * (1.0 / kap[i]);
    }
    kmath::log1p_pos(zz, ll);
    for (int i = 0; i < 4; ++i) ee[i] = -kap[i] * ll[i];
    kmath::exp_neg(ee, pp);
    for (int i = 0; i < 4; ++i) {
      const long double w = powl(1.0L + (long double)ss[i] / (long double)kap[i], -(long double)kap[i]);
      if (w > 1e-300L) {
        const double rel = (double)(fabsl((long double)pp[i] - w) / w);
        worst_p = std::fmax(worst_p, ulp_err(pp[i], w));
        worst_p_rel = std::fmax(worst_p_rel, rel);
        worst_p_scaled = std::fmax(worst_p_scaled, rel / (1.0 + std::fabs(ee[i])));
      }
    }
  }
  std::printf("exp_neg: %.2f ulp   log1p_pos: %.2f ulp   (1 + s/k)^-k: %.1f ulp, %.2e relative, %.2e relative / (1 + k log1p(s/k))\n",
              worst_e, worst_l, worst_p, worst_p_rel, worst_p_scaled);
  double x0[2] = {0.0, -1e300}, o0[2];
  kmath::exp_neg(x0, o0);
  double z0[2] = {0.0, 1e-300}, l0[2];
  kmath::log1p_pos(z0, l0);
  std::printf("exp(0) = %.17g exp(-1e300) = %g log1p(0) = %g log1p(1e-300) = %g\n", o0[0], o0[1], l0[0], l0[1]);
  return (worst_e <= 1.0 && worst_l <= 1.5 && o0[0] == 1.0 && o0[1] == 0.0 && l0[0] == 0.0) ? 0 : 1;
}
