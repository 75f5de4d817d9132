// O(N) kernels of the resident wake (wake.hip): mirror refresh after a host write, the staging of a time step's upload, the unit
// influences and chord-sum finisher of ludvm_wake_step, and the Euler finisher of the symmetric roll-up.  Non-template kernels:
// this header belongs to ONE translation unit (wake.hip).
#pragma once
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"

namespace ludvm {

// Rebuild the fp32 mirrors of every origin block that intersects [first, first + count) from the float64
// masters (after a host write): the block's two origins are re-taken (origin_index over the `stored` entries), so the
// whole block is refreshed.  Entries of a touched block that lie beyond the stored range are computed from whatever
// the master arrays hold there and are never read.  `limit` = allocated capacity.
__global__ void __launch_bounds__(kBlock)
refresh_mirrors(long long first, long long count, long long stored, long long limit, const double* x64, const double* z64,
                const double* g64, Mirrors m, float* g32) {
  const long long lo = (first >> kOriginShift) << kOriginShift;
  const long long i = lo + (long long)blockIdx.x * kBlock + threadIdx.x;
  long long hi = ((first + count + kOriginBlock - 1) >> kOriginShift) << kOriginShift;
  if (hi > limit) hi = limit;
  if (i >= hi) return;
  const long long b = i >> kOriginShift;
  const int p = (int)(i & 1);
  const long long oi = origin_index(b, p, stored);
  const float ox = (float)x64[oi], oz = (float)z64[oi];
  if ((i & (kOriginBlock - 1)) < 2) { m.cx[2 * b + p] = ox; m.cz[2 * b + p] = oz; }      // the first thread of each class
  store_mirrors(m, i, x64[i], z64[i], ox, oz);
  g32[i] = (float)g64[i];
}

// Velocity induced at nt points by n_unit unit-strength vortices (the new TEV / LEV of a time step,
// LUDVM.py:751, :926, :931), fp64: out[(k*2 + 0)*nt + p] = u, out[(k*2 + 1)*nt + p] = w.
__global__ void __launch_bounds__(kBlock)
unit_influence_f64(const double* xt, const double* zt, long long nt, const double* ux, const double* uz, int n_unit,
                   double vc4, double* out) {
  const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= nt * n_unit) return;
  const long long k = idx / nt, p = idx - k * nt;
  const double dx = xt[p] - ux[k];
  const double dz = zt[p] - uz[k];
  const double r2 = __builtin_fma(dz, dz, dx * dx);
  const double s = kInv2PiD * rsqrt_f64(__builtin_fma(r2, r2, vc4));
  out[(k * 2 + 0) * nt + p] = dz * s;
  out[(k * 2 + 1) * nt + p] = -dx * s;
}

// One kernel stages everything a time step uploads: `n_new` shed vortices appended at wake index n0 and
// `n_foil` bound vortices behind them (sources of the roll-up only), from one packed host->device copy
// pack = [new_x | new_z | new_g | foil_x | foil_z | foil_g]; float64 masters and fp32 mirrors are written.
// An origin class (block x index parity) whose first member lies inside the staged range takes its origin from the
// vortex staged there (the other classes keep the origin the last Euler finisher gave them).
__global__ void __launch_bounds__(kBlock)
stage_step_inputs(const double* pack, long long n0, int n_new, int n_foil, double* x64, double* z64, double* g64, Mirrors m,
                  float* g32) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n_new + n_foil) return;
  auto staged = [&](int q, double& x, double& z, double& g) {
    const double* base = q < n_new ? pack : pack + 3 * n_new;
    const int cnt = q < n_new ? n_new : n_foil;
    const int j = q < n_new ? q : q - n_new;
    x = base[j]; z = base[cnt + j]; g = base[2 * cnt + j];
  };
  double x, z, g;
  staged(k, x, z, g);
  const long long i = n0 + k;
  const long long b = i >> kOriginShift, cs = (b << kOriginShift) + (i & 1), slot = origin_slot(i);   // cs: first member of i's class
  float ox, oz;
  if (cs >= n0) {
    double bx, bz, bg;
    staged((int)(cs - n0), bx, bz, bg);
    ox = (float)bx; oz = (float)bz;
    if (i == cs) { m.cx[slot] = ox; m.cz[slot] = oz; }
    // a block opened by the last staged entry: give its still empty odd class a number too
    if (i == cs && (i & 1) == 0 && k + 1 == n_new + n_foil) { m.cx[slot + 1] = ox; m.cz[slot + 1] = oz; }
  } else {
    ox = m.cx[slot]; oz = m.cz[slot];
  }
  x64[i] = x; z64[i] = z; g64[i] = g;
  store_mirrors(m, i, x, z, ox, oz);
  g32[i] = (float)g;
}

// One launch for the three small jobs that follow a time step's roll-up (ludvm_wake_step):
//   (a) sum the fp64 wake->chord partial slabs                        -> out_sums[0 .. 2 nt)
//   (b) report the newest `tail` wake vortices and place the next time step's TEV and candidate LEV from
//       them (LUDVM.py:680-681, :797-800): one third of the way from the shedding edge to the newest TEV /
//       LEV.  geo = [te_x, te_z, le_x, le_z]; the newest TEV is vortex n - tail, the newest LEV vortex
//       n - 1 (when tail == 2 and lev_from_prev), else the candidate sits on the leading edge
//                                                       -> out_head = [tail x | tail z | tev_x, lev_x, tev_z, lev_z]
//   (c) velocities induced at the chord points by those two unit vortices (as unit_influence_f64)
//                                                                     -> out_sums[2 nt .. 6 nt)
// One WAVEFRONT per output column (k, p), k = 0: u / unit TEV, k = 1: w / unit LEV: lane l sums the
// splits s = l, l + 64, ... and the 64 partials are combined by a fixed shuffle tree (deterministic;
// a column of ~500 splits costs ~8 loads per lane instead of 500 dependent ones).  Lane 0 of the column
// also evaluates (c); the placements are recomputed by whoever needs them (a handful of flops).
// grid covers 2 * nt * 64 threads.
__global__ void __launch_bounds__(kBlock)
chord_finish_f64(const double* part, long long nt_pad, int nsplit, const double* direct_u, const double* xt, const double* zt,
                 long long nt, const double* x64, const double* z64, long long n, int tail, int lev_from_prev,
                 const double* geo, double vc4, double* out_head, double* out_sums) {
  const long long gtid = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long col = gtid >> 6;
  const int lane = threadIdx.x & 63;
  const double tex = geo[0], tez = geo[1], lex = geo[2], lez = geo[3];
  const long long it = n - tail;
  double ux[2], uz[2];
  ux[0] = tex + (x64[it] - tex) / 3;
  uz[0] = tez + (z64[it] - tez) / 3;
  if (lev_from_prev && tail == 2) {
    ux[1] = lex + (x64[n - 1] - lex) / 3;
    uz[1] = lez + (z64[n - 1] - lez) / 3;
  } else {
    ux[1] = lex;
    uz[1] = lez;
  }
  if (gtid == 0) {
    for (int t = 0; t < tail; ++t) { out_head[t] = x64[n - tail + t]; out_head[tail + t] = z64[n - tail + t]; }
    double* unit = out_head + 2 * tail;
    unit[0] = ux[0]; unit[1] = ux[1]; unit[2] = uz[0]; unit[3] = uz[1];
  }
  if (col >= 2 * nt) return;   // whole wavefronts leave together
  const long long k = col / nt, p = col - k * nt;
  // (a) component k (0: u, 1: w) of the wake sum at chord point p
  double acc = 0.0;
  if (part != nullptr) {
    const double* c0 = part + k * nt_pad + p;
    for (int sidx = lane; sidx < nsplit; sidx += 64) acc += c0[(long long)sidx * 2 * nt_pad];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  } else if (nsplit == 1) {          // one split: the pair kernel wrote u | w directly
    acc = direct_u[k * nt_pad + p];
  }
  if (lane != 0) return;
  out_sums[k * nt + p] = acc;
  // (c) unit vortex k at chord point p
  const double dx = xt[p] - ux[k];
  const double dz = zt[p] - uz[k];
  const double r2 = __builtin_fma(dz, dz, dx * dx);
  const double s = kInv2PiD * rsqrt_f64(__builtin_fma(r2, r2, vc4));
  out_sums[2 * nt + (k * 2 + 0) * nt + p] = dz * s;
  out_sums[2 * nt + (k * 2 + 1) * nt + p] = -dx * s;
}

// Resident-wake Euler step from the symmetric kernel's raw sums, plus the velocity induced by the nfoil bound
// vortices staged behind the wake at index nt: float64 update of the master copy, refresh of the fp32 mirrors
// (LUDVM.py:1108-1127).  One workgroup = one origin block (see finish_wake_advect).
__global__ void __launch_bounds__(kFinBlock)
finish_wake_advect_sym(const long long* acc_u, const long long* acc_w, const SymScale* sc, const long long* bad, long long nt,
                       int nfoil, float vc4, double dt, double* x64, double* z64, Mirrors m, const float* g32, double* u_out,
                       double* w_out, const long long* n_dev = nullptr, TailDuty td = TailDuty{}) {
  __shared__ float org[4];
  if (n_dev) nt = *n_dev;          // device-resident march: the wake size lives on the device
  tail_duty_block0(td, nt);
  const long long i = (long long)blockIdx.x * kFinBlock + threadIdx.x;
  const bool on = i < nt;
  const double xo = on ? x64[i] : 0.0, zo = on ? z64[i] : 0.0;
  float fu, fw;
  staged_sources_on(on, xo, zo, x64, z64, g32, nt, nfoil, vc4, fu, fw);
  double xn = 0.0, zn = 0.0;
  if (on) {
    const float s = (float)kInv2PiD;
    const float su = (fx_read(acc_u, i, sc, bad) + fu) * s, sw = -(fx_read(acc_w, i, sc, bad) + fw) * s;
    if (u_out) { u_out[i] = (double)su; w_out[i] = (double)sw; }
    xn = xo + dt * (double)su;
    zn = zo + dt * (double)sw;
    publish_origins(m, i, nt, xn, zn, org);
  }
  __syncthreads();
  if (!on) return;
  x64[i] = xn;
  z64[i] = zn;
  store_mirrors(m, i, xn, zn, org[i & 1], org[2 + (i & 1)]);
  tail_duty(td, i, nt, xn, zn);
}

}  // namespace ludvm
