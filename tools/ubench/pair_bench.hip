// Standalone A/B harness for the pair kernels (not part of the product path).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../ludvm_amd/csrc pair_bench.hip -o pair_bench
#include "pair_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace ludvm;

struct Variant { const char* name; void (*fn)(PairArgs<float>); int tpl; int tile; };

int main(int argc, char** argv) {
  long long N = argc > 1 ? atoll(argv[1]) : 262144;
  int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::mt19937_64 rng(1234);
  std::uniform_real_distribution<double> ux(-10, 0), uz(-2, 2);
  std::normal_distribution<double> ng(0, 1);
  std::vector<float> x(N), z(N), g(N);
  for (long long i = 0; i < N; ++i) { x[i] = (float)ux(rng); z[i] = (float)uz(rng); g[i] = (float)(ng(rng) / N); }
  float *dx, *dz, *dg, *du, *dw, *part;
  CK(hipMalloc(&dx, N * 4)); CK(hipMalloc(&dz, N * 4)); CK(hipMalloc(&dg, N * 4));
  CK(hipMalloc(&du, N * 4)); CK(hipMalloc(&dw, N * 4));
  const int max_split = 64;
  CK(hipMalloc(&part, (size_t)max_split * 2 * N * 4));
  CK(hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dz, z.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g.data(), N * 4, hipMemcpyHostToDevice));
  const float vc = 0.065f;

  // CPU double reference on a handful of targets
  const int nchk = 16;
  std::vector<double> ru(nchk), rw(nchk);
  std::vector<long long> idx(nchk);
  for (int c = 0; c < nchk; ++c) {
    long long p = (N / nchk) * c + c; idx[c] = p;
    double su = 0, sw = 0, v4 = pow((double)vc, 4);
    for (long long j = 0; j < N; ++j) {
      double ddx = (double)x[p] - x[j], ddz = (double)z[p] - z[j];
      double r2 = ddx * ddx + ddz * ddz;
      double k = g[j] / (2 * M_PI * sqrt(r2 * r2 + v4));
      su += ddz * k; sw -= ddx * k;
    }
    ru[c] = su; rw[c] = sw;
  }

  Variant vars[] = {
    {"packed T1 tile1024", pair_f32_packed<1, 1024>, 1, 1024},
    {"packed T2 tile1024", pair_f32_packed<2, 1024>, 2, 1024},
    {"packed T4 tile1024", pair_f32_packed<4, 1024>, 4, 1024},
    {"packed T2 tile2048", pair_f32_packed<2, 2048>, 2, 2048},
    {"scalar T1 tile1024", pair_f32_scalar<1, 1024>, 1, 1024},
    {"scalar T2 tile1024", pair_f32_scalar<2, 1024>, 2, 1024},
    {"scalar T4 tile1024", pair_f32_scalar<4, 1024>, 4, 1024},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& v : vars) {
    for (int nsplit : {1, 4, 16}) {
      PairArgs<float> a{};
      a.xs = dx; a.zs = dz; a.gs = dg; a.xt = dx; a.zt = dz; a.u = du; a.w = dw; a.part = part;
      a.ns = N; a.nt = N; a.nt_pad = N;
      long long chunk = (N + nsplit - 1) / nsplit;
      chunk = (chunk + v.tile - 1) / v.tile * v.tile;
      a.chunk = chunk;
      int ns_eff = (int)((N + chunk - 1) / chunk);
      a.vc4 = vc * vc * vc * vc; a.out_scale = kInv2Pi;
      dim3 grid((unsigned)((N + (long long)kBlock * v.tpl - 1) / ((long long)kBlock * v.tpl)), ns_eff);
      auto launch = [&]() {
        hipLaunchKernelGGL(v.fn, grid, dim3(kBlock), 0, 0, a);
        if (ns_eff > 1)
          hipLaunchKernelGGL(reduce_splits<float>, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                             part, N, N, ns_eff, du, dw);
      };
      CK(hipMemset(du, 0, N * 4)); CK(hipMemset(dw, 0, N * 4));
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
      for (int c = 0; c < nchk; ++c) {
        maxerr = std::max(maxerr, std::max(fabs(hu[idx[c]] - ru[c]), fabs(hw[idx[c]] - rw[c])));
        maxref = std::max(maxref, std::max(fabs(ru[c]), fabs(rw[c])));
      }
      double pairs = (double)N * N;
      double pps = pairs / (ms[ms.size() / 2] * 1e-3);
      printf("%-22s split %2d grid %5ux%-2u  med %.3f ms  min %.3f ms  %.3e pairs/s  %.1f TFLOP/s (%.1f%% of 157.3)  relerr %.2e\n",
             v.name, ns_eff, grid.x, grid.y, ms[ms.size() / 2], ms[0], pps, pps * 13 / 1e12, pps * 13 / 157.3e12 * 100,
             maxerr / maxref);
    }
  }
  return 0;
}
