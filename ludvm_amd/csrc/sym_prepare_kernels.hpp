// The fixed-point scale of a symmetric launch (sum|Gamma| -> SymScale) and the fixed-point probe.  Non-template kernels: this
// header belongs to ONE translation unit (launch.hip).
#pragma once
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"

namespace ludvm {

// measurement / test probe (ludvm_fixed_point_probe): out[i] = the integer fx_add would add for v[i]
__global__ void __launch_bounds__(kBlock) fx_probe(const float* v, long long n, float scale, long long* out) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (long long)fx_units(v[i], scale);
}

__global__ void __launch_bounds__(kPrepBlock)
sym_prepare(const float* g, long long n, double vc4, SymScale* out, long long* bad, double* partial) {
  if (gridDim.x == 1) {
    const double tot = block_abs_sum(g, 0, n);
    if (threadIdx.x == 0) { *bad = 0; sym_scale_from_sum(tot, vc4, out, bad); }
    return;
  }
  const long long first = (long long)blockIdx.x * kPrepChunk;
  const long long cnt = n - first < kPrepChunk ? n - first : kPrepChunk;
  const double tot = block_abs_sum(g, first, cnt);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(64)
sym_prepare_final(const double* partial, int nparts, double vc4, SymScale* out, long long* bad) {
  if (threadIdx.x != 0) return;
  double tot = 0.0;
  for (int k = 0; k < nparts; ++k) tot += partial[k];
  *bad = 0;
  sym_scale_from_sum(tot, vc4, out, bad);
}

}  // namespace ludvm
