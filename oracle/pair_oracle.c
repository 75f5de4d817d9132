/*
 * pair_oracle.c -- plain C float64 restatement of the reference's pair sum.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/ludvm_oracle.py).  Follows LUDVM.induced_velocity,
 * /root/reference/LUDVM.py:549-570, term by term: the denominator 2*pi*sqrt((dx^2+dz^2)^2 + vc^4) is
 * evaluated for Ku and again for Kw (:565-566), u += G*Ku, w += -G*Kw (:568-569).  The only liberty is
 * the order of the row sum (sequential here, NumPy pairwise there), worth <= 1e-13 relative.
 * Used where the NumPy restatement is too slow: sampled-target checks of N = 1e6 launches on the GPU
 * box.  Rows (targets) are independent, so they are split over OpenMP threads when built with
 * -fopenmp.  Build: make -C oracle
 */
#include <math.h>
#include <stddef.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void pair_oracle_f64(const double* g, const double* xw, const double* zw, size_t nw, const double* xp,
                     const double* zp, size_t np, double v_core, double* u, double* w) {
  const double vc4 = v_core * v_core * v_core * v_core;
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)np; ++p) {
    double su = 0.0, sw = 0.0;
    for (size_t k = 0; k < nw; ++k) {
      const double xd = xp[p] - xw[k];
      const double zd = zp[p] - zw[k];
      const double r2a = xd * xd + zd * zd;
      const double ku = zd / (2 * M_PI * sqrt(r2a * r2a + vc4));
      const double r2b = xd * xd + zd * zd;
      const double kw = xd / (2 * M_PI * sqrt(r2b * r2b + vc4));
      su += g[k] * ku;
      sw += -g[k] * kw;
    }
    u[p] = su;
    w[p] = sw;
  }
}

/* Rows are independent: the thread count changes nothing but the time.  A launcher that starts several ranks per node
 * pins OMP_NUM_THREADS to 1; the one rank that runs a sampled check while the others wait may ask for more. */
void pair_oracle_set_threads(int n) {
#ifdef _OPENMP
  extern void omp_set_num_threads(int);
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int pair_oracle_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
