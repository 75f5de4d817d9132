// Harness for the symmetric self-interaction kernel (not part of the product path).
#include "pair_sym_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace ludvm;

template <int T>
void run(const char* name, long long N, int ysplit, int reps, float* dx, float* dz, float* dg, float* au, float* aw, float* du,
         float* dw, const std::vector<long long>& idx, const std::vector<double>& ru, const std::vector<double>& rw) {
  SymArgs a{};
  a.x = dx; a.z = dz; a.g = dg; a.n = N;
  const long long W = 64LL * T;
  a.ntiles = (N + W - 1) / W;
  a.dmax = (a.ntiles - 1) / 2;
  a.i_first = 0; a.i_count = a.ntiles;
  a.ysplit = ysplit;
  a.acc_u = au; a.acc_w = aw;
  const float vc = 0.065f;
  a.vc4 = vc * vc * vc * vc;
  const long long waves = a.ntiles * ysplit;
  dim3 grid((unsigned)((waves + 3) / 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    CK(hipMemsetAsync(au, 0, N * 4)); CK(hipMemsetAsync(aw, 0, N * 4));
    hipLaunchKernelGGL((pair_sym_f32<T, false>), grid, dim3(kBlock), 0, 0, a);
    hipLaunchKernelGGL(finish_sym, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, au, aw, N, du, dw);
  };
  launch(); CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  std::vector<float> hu(N), hw(N);
  CK(hipMemcpy(hu.data(), du, N * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hw.data(), dw, N * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  for (size_t c = 0; c < idx.size(); ++c) {
    maxerr = std::max(maxerr, std::max(fabs(hu[idx[c]] - ru[c]), fabs(hw[idx[c]] - rw[c])));
    maxref = std::max(maxref, std::max(fabs(ru[c]), fabs(rw[c])));
  }
  double pps = (double)N * N / (ms[ms.size() / 2] * 1e-3);
  printf("%-10s N %lld ysplit %2d grid %6u  med %.3f ms  %.3e ordered pairs/s  %.1f%% of 157.3 TF (13 flop/pair)  relerr %.2e\n",
         name, N, ysplit, grid.x, ms[ms.size() / 2], pps, pps * 13 / 157.3e12 * 100, maxerr / maxref);
}

int main(int argc, char** argv) {
  long long N = argc > 1 ? atoll(argv[1]) : 262144;
  int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::mt19937_64 rng(1234);
  std::uniform_real_distribution<double> ux(-10, 0), uz(-2, 2);
  std::normal_distribution<double> ng(0, 1);
  std::vector<float> x(N), z(N), g(N);
  for (long long i = 0; i < N; ++i) { x[i] = (float)ux(rng); z[i] = (float)uz(rng); g[i] = (float)(ng(rng) / N); }
  float *dx, *dz, *dg, *du, *dw, *au, *aw;
  CK(hipMalloc(&dx, N * 4)); CK(hipMalloc(&dz, N * 4)); CK(hipMalloc(&dg, N * 4));
  CK(hipMalloc(&du, N * 4)); CK(hipMalloc(&dw, N * 4)); CK(hipMalloc(&au, N * 4)); CK(hipMalloc(&aw, N * 4));
  CK(hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dz, z.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g.data(), N * 4, hipMemcpyHostToDevice));
  const int nchk = 48;
  std::vector<double> ru(nchk), rw(nchk);
  std::vector<long long> idx(nchk);
  for (int c = 0; c < nchk; ++c) {
    long long p = (c < 8) ? c * 37 : (c >= nchk - 8 ? N - 1 - (nchk - 1 - c) * 29 : (N / nchk) * c + c);
    idx[c] = p;
    double su = 0, sw = 0, v4 = pow(0.065, 4);
    for (long long j = 0; j < N; ++j) {
      double ddx = (double)x[p] - x[j], ddz = (double)z[p] - z[j];
      double r2 = ddx * ddx + ddz * ddz;
      double k = g[j] / (2 * M_PI * sqrt(r2 * r2 + v4));
      su += ddz * k; sw -= ddx * k;
    }
    ru[c] = su; rw[c] = sw;
  }
  auto ys_for = [&](int T) {
    long long ntiles = (N + 64LL * T - 1) / (64LL * T);
    long long dtot = (ntiles - 1) / 2 + ((ntiles % 2 == 0 && ntiles > 1) ? 1 : 0);
    long long ys = (65536 + ntiles - 1) / ntiles;
    ys = std::max<long long>(1, std::min<long long>(std::min<long long>(ys, 64), std::max<long long>(dtot, 1)));
    return (int)ys;
  };
  for (int rep = 0; rep < 3; ++rep) {
    run<4>("sym T4", N, ys_for(4), reps, dx, dz, dg, au, aw, du, dw, idx, ru, rw);
    run<8>("sym T8", N, ys_for(8), reps, dx, dz, dg, au, aw, du, dw, idx, ru, rw);
  }
  return 0;
}
