// O(N) kernels of the stateless entry points (induce.hip): staging conversions of the packed host block, the summed results of
// the symmetric kernel, and the fp32 Euler steps of the multi-GPU shard step.  Non-template kernels: this header belongs to ONE
// translation unit (induce.hip); the templates and device helpers they use come from pair_kernels.hpp / pair_sym_kernels.hpp.
#pragma once
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"

namespace ludvm {

// fp32 device SoA Euler step: x_out[i] = x[t_first + i] + dt * u_i  (LUDVM.py:1108-1109).
// src_u/src_w are either the partial slabs (nsplit > 1) or the direct results (nsplit == 1,
// nt_pad == stride between u and w rows is irrelevant then: pass part = u, and w separately).
__global__ void __launch_bounds__(kBlock)
finish_advect_f32(const float* part, long long nt, long long nt_pad, int nsplit, const float* x, const float* z,
                  long long t_first, float dt, float* x_out, float* z_out) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nt) return;
  float su, sw;
  sum_splits(part, i, nt_pad, nsplit, su, sw);
  x_out[i] = __builtin_fmaf(dt, su, x[t_first + i]);
  z_out[i] = __builtin_fmaf(dt, sw, z[t_first + i]);
}

// Small stateless calls: the five float64 input arrays arrive in one packed upload, in = xs[ns] | zs[ns] | gs[ns] |
// xt[nt] | zt[nt]; one launch splits them into the fp32 (hi, lo) arrays the kernels read.
__global__ void __launch_bounds__(kBlock)
cvt_packed_inputs(const double* in, long long ns, long long nt, float* xs, float* xsl, float* zs, float* zsl, float* gs,
                  float* xt, float* xtl, float* zt, float* ztl) {
  const long long k = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (k >= 3 * ns + 2 * nt) return;
  const double v = in[k];
  float h, l;
  split_hilo(v, h, l);
  if (k < ns) { xs[k] = h; xsl[k] = l; }
  else if (k < 2 * ns) { zs[k - ns] = h; zsl[k - ns] = l; }
  else if (k < 3 * ns) { gs[k - 2 * ns] = h; }
  else if (k < 3 * ns + nt) { xt[k - 3 * ns] = h; xtl[k - 3 * ns] = l; }
  else { zt[k - 3 * ns - nt] = h; ztl[k - 3 * ns - nt] = l; }
}

// The same packed block as local-origin fp32: offsets from the origins of each array's own origin classes (256-element
// block x index parity; origin = fp32 value of the class's middle element).  nt = 0 when the targets are the sources
// themselves.
__global__ void __launch_bounds__(kBlock)
cvt_packed_inputs_local(const double* in, long long ns, long long nt, float* xs, float* zs, float* gs, float* sox, float* soz,
                        float* xt, float* zt, float* tox, float* toz) {
  const long long k = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (k >= 3 * ns + 2 * nt) return;
  const double v = in[k];
  auto local = [&](long long base, long long i, long long len, float* off, float* org) {
    const long long b = i >> kOriginShift;
    const int p = (int)(i & 1);
    // (a class whose middle member is not a number -- a NaN target poisons only itself in the reference's sum -- takes 0)
    const float o_raw = (float)in[base + origin_index(b, p, len)];
    const float o = __builtin_fabsf(o_raw) < __builtin_inff() ? o_raw : 0.0f;
    if ((i & (kOriginBlock - 1)) < 2) org[2 * b + p] = o;
    // (a block that holds a single element: its odd class has no member to write the record, which the kernels still read)
    if ((i & (kOriginBlock - 1)) == 0 && i + 1 >= len) org[2 * b + 1] = o;
    off[i] = (float)(v - (double)o);
  };
  if (k < ns) local(0, k, ns, xs, sox);
  else if (k < 2 * ns) local(ns, k - ns, ns, zs, soz);
  else if (k < 3 * ns) gs[k - 2 * ns] = (float)v;
  else if (k < 3 * ns + nt) local(3 * ns, k - 3 * ns, nt, xt, tox);
  else local(3 * ns + nt, k - 3 * ns - nt, nt, zt, toz);
}

// ... and the two fp32 results leave as one float64 block out = u[nt] | w[nt].
__global__ void __launch_bounds__(kBlock)
cvt_packed_outputs(const float* u, const float* w, double* out, long long nt) {
  const long long k = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (k >= 2 * nt) return;
  out[k] = k < nt ? (double)u[k] : (double)w[k - nt];
}

// acc -> velocities
__global__ void __launch_bounds__(kBlock)
finish_sym(const long long* acc_u, const long long* acc_w, const SymScale* sc, const long long* bad, long long n, float* u,
           float* w) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float s = (float)kInv2PiD;
  u[i] = fx_read(acc_u, i, sc, bad) * s;
  w[i] = -fx_read(acc_w, i, sc, bad) * s;
}

// raw sums of targets [t_first, t_first + nt) (sum_u[i], sum_w[i] belong to target t_first + i) -> Euler step
__global__ void __launch_bounds__(kBlock)
finish_sym_advect(const long long* sum_u, const long long* sum_w, const SymScale* sc, const long long* bad, const float* x,
                  const float* z, long long t_first, long long nt, float dt, float* x_out, float* z_out) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nt) return;
  const float s = (float)kInv2PiD;
  x_out[i] = __builtin_fmaf(dt, fx_read(sum_u, i, sc, bad) * s, x[t_first + i]);
  z_out[i] = __builtin_fmaf(dt, -fx_read(sum_w, i, sc, bad) * s, z[t_first + i]);
}

}  // namespace ludvm
