// Symmetric ("Newton's third law") Vatistas-core Biot-Savart kernel for SELF-interaction launches
// (targets == sources: wake roll-up LUDVM.py:1105-1124, BASELINE configs 3 and 4) on gfx950.
//
// The pair kernel is antisymmetric, K(i->j) = -K(j->i): dx, dz, r^2, q and -- the expensive part --
// rsq(q) are shared by the ordered pairs (i,j) and (j,i).  Direct evaluation costs 8 full-rate VALU
// ops + 1 v_rsq_f32 per ordered pair (24 issue cycles per 64 lanes on CDNA4); evaluating each
// UNORDERED pair once costs 11 ops + 1 rsq per TWO ordered pairs (15 cycles per ordered pair).
//
// Wave-autonomous systolic scheme (no LDS, no barriers, no cross-lane reductions):
//   * vortices are cut into tiles of W = 64*T; a wavefront owns one tile I: lane l keeps T targets
//     (x, z, Gamma, accumulators) in registers for its whole life;
//   * for each partner tile J the lane loads T vortices of J the same way, with their own
//     accumulators; lane l then evaluates its T x T pairs IN REGISTERS, updating both sides, and the
//     J set (x, z, Gamma and its accumulators) is rotated one lane with v_mov_b32_dpp wave_rol:1;
//     after 64 rotations every i of I has met every j of J and the J accumulators are back home;
//   * tile pairs are enumerated cyclically, J = I + d (mod NT), d = 1 .. (NT-1)/2, so every unordered
//     tile pair is visited once and every wave has the same amount of work; d = 0 (the diagonal tile,
//     which holds the self pairs) is evaluated with the ordered formula.  Because the work of a tile I
//     depends on I alone, a GPU that owns a contiguous block of I tiles does exactly 1/G of the job
//     (multi-GPU: the raw sums are then reduce-scattered, see ludvm_amd/sharded.py);
//   * results are accumulated with float atomics into acc_u / acc_w (zeroed by the caller) and turned
//     into velocities (or an Euler step) by a finisher.  The summation order of the atomics is not
//     fixed: results are reproducible to rounding, not bitwise (the direct kernel is bitwise).
#pragma once
#include <hip/hip_runtime.h>
#include "pair_kernels.hpp"

namespace ludvm {

struct SymArgs {
  const float* x; const float* z; const float* g;  // N vortices (device)
  const float* xl; const float* zl;                // lo parts of the positions (HILO kernels only)
  long long n;
  long long ntiles;      // ceil(n / (64*T))
  long long dmax;        // floor((ntiles-1)/2): symmetric offsets 1..dmax (+ ntiles/2 when even)
  long long i_first;     // this launch owns I tiles [i_first, i_first + i_count): one GPU's block of the
  long long i_count;     //   global tile ring (multi-GPU), or all tiles
  int ysplit;            // number of d-chunks (gridDim.x = ceil(i_count*ysplit*rsplit / 4))
  int rsplit;            // 1, 2 or 4: the 64 rotation steps of a tile pair are shared by this many waves
  float* acc_u; float* acc_w;  // raw sums: u = acc_u/(2 pi), w = -acc_w/(2 pi)
  float vc4;
  // Device-resident march (single GPU, all tiles): n is read from memory and ntiles / dmax / i_count follow
  // from it; the grid and ysplit come from the host's upper bound, surplus waves leave at once.
  const long long* n_dev;
};

// lane l receives the value of lane l+1 (wrapping): data moves one lane down
__device__ __forceinline__ float dpp_rol1(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x134, 0xf, 0xf, false));
}
__device__ __forceinline__ f32x2 dpp_rol1(f32x2 v) { return (f32x2){dpp_rol1(v.x), dpp_rol1(v.y)}; }

// Slab helpers: a wave's LDS slab holds T floats per home lane and component, as T/4 planes of
// [64 home lanes][4 floats]: every ds_read_b128 / ds_write_b128 of a wave then covers 64 consecutive
// 16-byte slots (conflict-free); `home4` is 4 * home lane.
template <int T>
__device__ __forceinline__ void slab_store(float* l, int home4, const float (&v)[T]) {
#pragma unroll
  for (int q = 0; q < T / 4; ++q)
    *reinterpret_cast<f32x4*>(&l[q * 256 + home4]) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
template <int T>
__device__ __forceinline__ void slab_load(const float* l, int home4, f32x2 (&out)[T / 2]) {
#pragma unroll
  for (int q = 0; q < T / 4; ++q) {
    const f32x4 V = *reinterpret_cast<const f32x4*>(&l[q * 256 + home4]);
    out[2 * q] = (f32x2){V.x, V.y};
    out[2 * q + 1] = (f32x2){V.z, V.w};
  }
}

// T vortices per lane on both sides (tile = 64*T).  Per rotation step a lane evaluates T*T unordered
// pairs with 5.5*T*T packed ops + T*T v_rsq_f32; the J tile's (x, z, Gamma) sit in a wave-private LDS slab
// and are read with a per-lane rotating address (T/4 ds_read_b128 per component, off the VALU pipe); only
// the 2*T J-accumulator registers travel between lanes (v_mov_b32_dpp, 4 issue cycles each on gfx950).
// HILO: positions are hi+lo fp32 pairs, dx = (xh_i - xh_j) + (xl_i - xl_j), +4 packed ops
// per two unordered pairs; everything after the difference is plain fp32.
template <int T, bool HILO = false>
__global__ void __launch_bounds__(kBlock)
pair_sym_f32(SymArgs a) {
  static_assert(T == 4 || T == 8, "T vortices per lane, read as T/4 ds_read_b128 per component");
  if (a.n_dev) {
    a.n = *a.n_dev;
    a.ntiles = (a.n + 64LL * T - 1) / (64LL * T);
    a.dmax = (a.ntiles - 1) / 2;
    a.i_count = a.ntiles;
  }
  constexpr int H = T / 2;
  constexpr int kWaves = kBlock / 64;
  constexpr int kComp = HILO ? 5 : 3;
  __shared__ __attribute__((aligned(16))) float slab[kWaves][kComp][64 * T];

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const long long wid = (long long)blockIdx.x * kWaves + wv;
  if (wid >= a.i_count * a.ysplit * a.rsplit) return;   // whole waves leave together; no block-wide barrier is used
  const long long I = a.i_first + wid % a.i_count;
  const int yr = (int)(wid / a.i_count);
  const int y = yr / a.rsplit;
  // Mid-size launches have too few tile pairs to keep every SIMD busy to the end, so a tile pair's 64 rotation
  // steps can be shared by rsplit waves: this one does steps [k_lo, k_hi).  A J accumulator set that starts in
  // lane l at step k_lo belongs to home lane (l + k_lo) and, one lane per step, sits in lane (home - k_hi) after
  // the last step -- it is added to its vortices from there (the sums are atomics anyway).
  const int k_lo = (yr % a.rsplit) * (64 / a.rsplit);
  const int k_hi = k_lo + 64 / a.rsplit;
  const long long W = 64LL * T;
  float* const lx = slab[wv][0];
  float* const lz = slab[wv][1];
  float* const lg = slab[wv][2];
  float* const lxl = slab[wv][HILO ? 3 : 0];
  float* const lzl = slab[wv][HILO ? 4 : 1];

  const bool even = (a.ntiles % 2 == 0) && a.ntiles > 1;
  const long long dtot = a.dmax + (even ? 1 : 0);
  const long long per = (dtot + a.ysplit - 1) / a.ysplit;
  const long long d_lo = 1 + (long long)y * per;
  long long d_hi = d_lo + per;  // exclusive
  if (d_hi > dtot + 1) d_hi = dtot + 1;

  // my targets (duplicated into register pairs: the packed ops pair two SOURCES against one target)
  f32x2 xp[T], zp[T], gp[T], au[T], aw[T], xpl[T], zpl[T];
  float x0[T], z0[T], g0[T], xl0[T], zl0[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const long long i = I * W + lane + 64LL * t;
    const bool ok = i < a.n;
    x0[t] = ok ? a.x[i] : kPadPosF; z0[t] = ok ? a.z[i] : kPadPosF; g0[t] = ok ? a.g[i] : 0.0f;
    xl0[t] = (HILO && ok) ? a.xl[i] : 0.0f; zl0[t] = (HILO && ok) ? a.zl[i] : 0.0f;
    xp[t] = (f32x2){x0[t], x0[t]}; zp[t] = (f32x2){z0[t], z0[t]}; gp[t] = (f32x2){g0[t], g0[t]};
    xpl[t] = (f32x2){xl0[t], xl0[t]}; zpl[t] = (f32x2){zl0[t], zl0[t]};
    au[t] = (f32x2){0.f, 0.f}; aw[t] = (f32x2){0.f, 0.f};
  }
  const f32x2 vc4 = {a.vc4, a.vc4};

  // ---- diagonal tile: ordered evaluation, i-side only (contains the self pairs) ----------------
  if (y == 0) {
    slab_store<T>(lx, lane * 4, x0); slab_store<T>(lz, lane * 4, z0); slab_store<T>(lg, lane * 4, g0);
    if (HILO) { slab_store<T>(lxl, lane * 4, xl0); slab_store<T>(lzl, lane * 4, zl0); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = k_lo; k < k_hi; ++k) {
      const int pos = ((lane + k) & 63) * 4;
      f32x2 xj[H], zj[H], gj[H], xjl[H], zjl[H];
      slab_load<T>(lx, pos, xj); slab_load<T>(lz, pos, zj); slab_load<T>(lg, pos, gj);
      if (HILO) { slab_load<T>(lxl, pos, xjl); slab_load<T>(lzl, pos, zjl); }
#pragma unroll
      for (int m = 0; m < H; ++m) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          f32x2 dx = xp[t] - xj[m];
          f32x2 dz = zp[t] - zj[m];
          if (HILO) { dx = dx + (xpl[t] - xjl[m]); dz = dz + (zpl[t] - zjl[m]); }
          f32x2 r2 = dx * dx;
          r2 = __builtin_elementwise_fma(dz, dz, r2);
          const f32x2 q = __builtin_elementwise_fma(r2, r2, vc4);
          f32x2 s = {__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)};
          s = s * gj[m];
          au[t] = __builtin_elementwise_fma(dz, s, au[t]);
          aw[t] = __builtin_elementwise_fma(dx, s, aw[t]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- off-diagonal tiles: each unordered pair once, both sides accumulated --------------------
  for (long long d = d_lo; d < d_hi; ++d) {
    if (even && d == dtot && I >= a.ntiles / 2) break;  // the half-way offset pairs each tile twice
    long long J = I + d;
    if (J >= a.ntiles) J -= a.ntiles;
    {
      float x[T], z[T], g[T], xl[T], zl[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const long long j = J * W + lane + 64LL * t;
        const bool ok = j < a.n;
        x[t] = ok ? a.x[j] : kPadPosF; z[t] = ok ? a.z[j] : kPadPosF; g[t] = ok ? a.g[j] : 0.0f;
        xl[t] = (HILO && ok) ? a.xl[j] : 0.0f; zl[t] = (HILO && ok) ? a.zl[j] : 0.0f;
      }
      slab_store<T>(lx, lane * 4, x); slab_store<T>(lz, lane * 4, z); slab_store<T>(lg, lane * 4, g);
      if (HILO) { slab_store<T>(lxl, lane * 4, xl); slab_store<T>(lzl, lane * 4, zl); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x2 bu[H], bw[H];
#pragma unroll
    for (int m = 0; m < H; ++m) { bu[m] = (f32x2){0.f, 0.f}; bw[m] = (f32x2){0.f, 0.f}; }

    // (issuing the reads of step k+1 ahead of the arithmetic of step k measured no gain: with 4-5
    // waves per SIMD the LDS latency is already covered)
    for (int k = k_lo; k < k_hi; ++k) {
      // at step k this lane holds the accumulators of the J vortices whose home lane is (lane + k) % 64
      const int pos = ((lane + k) & 63) * 4;
      f32x2 xj[H], zj[H], gj[H], xjl[H], zjl[H];
      slab_load<T>(lx, pos, xj); slab_load<T>(lz, pos, zj); slab_load<T>(lg, pos, gj);
      if (HILO) { slab_load<T>(lxl, pos, xjl); slab_load<T>(lzl, pos, zjl); }
#pragma unroll
      for (int m = 0; m < H; ++m) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          f32x2 dx = xp[t] - xj[m];
          f32x2 dz = zp[t] - zj[m];
          if (HILO) { dx = dx + (xpl[t] - xjl[m]); dz = dz + (zpl[t] - zjl[m]); }
          f32x2 r2 = dx * dx;
          r2 = __builtin_elementwise_fma(dz, dz, r2);
          const f32x2 q = __builtin_elementwise_fma(r2, r2, vc4);
          const f32x2 s = {__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)};
          const f32x2 sj = s * gj[m];      // strength of j acting on i
          const f32x2 si = s * gp[t];      // strength of i acting on j
          au[t] = __builtin_elementwise_fma(dz, sj, au[t]);
          aw[t] = __builtin_elementwise_fma(dx, sj, aw[t]);
          bu[m] = __builtin_elementwise_fma(dz, si, bu[m]);
          bw[m] = __builtin_elementwise_fma(dx, si, bw[m]);
        }
      }
      // hand the J accumulators to the lane that meets the same J vortices next step (lane - 1)
#pragma unroll
      for (int m = 0; m < H; ++m) { bu[m] = dpp_rol1(bu[m]); bw[m] = dpp_rol1(bw[m]); }
    }
    // after step k_hi - 1 and its rotation this lane holds the set of home lane (lane + k_hi) % 64 (with all 64
    // steps done: its own); j feels the opposite of what i feels.
    // home lane h holds vortices J*W + h + 64*t as packed elements t = 0..T-1
    const int home = (lane + k_hi) & 63;
#pragma unroll
    for (int m = 0; m < H; ++m) {
      const long long j0 = J * W + home + 64LL * (2 * m), j1 = j0 + 64;
      if (j0 < a.n) { atomicAdd(&a.acc_u[j0], -bu[m].x); atomicAdd(&a.acc_w[j0], -bw[m].x); }
      if (j1 < a.n) { atomicAdd(&a.acc_u[j1], -bu[m].y); atomicAdd(&a.acc_w[j1], -bw[m].y); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the slab is rewritten by the next tile
  }

#pragma unroll
  for (int t = 0; t < T; ++t) {
    const long long i = I * W + lane + 64LL * t;
    if (i < a.n) { atomicAdd(&a.acc_u[i], au[t].x + au[t].y); atomicAdd(&a.acc_w[i], aw[t].x + aw[t].y); }
  }
}

// acc -> velocities
__global__ void __launch_bounds__(kBlock)
finish_sym(const float* acc_u, const float* acc_w, long long n, float* u, float* w) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float s = (float)kInv2PiD;
  u[i] = acc_u[i] * s;
  w[i] = -acc_w[i] * s;
}

// raw sums of targets [t_first, t_first + nt) (sum_u[i], sum_w[i] belong to target t_first + i) -> Euler step
__global__ void __launch_bounds__(kBlock)
finish_sym_advect(const float* sum_u, const float* sum_w, const float* x, const float* z, long long t_first, long long nt,
                  float dt, float* x_out, float* z_out) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nt) return;
  const float s = (float)kInv2PiD;
  x_out[i] = __builtin_fmaf(dt, sum_u[i] * s, x[t_first + i]);
  z_out[i] = __builtin_fmaf(dt, -sum_w[i] * s, z[t_first + i]);
}

// Resident-wake Euler step from the symmetric kernel's raw sums, plus the velocity induced by the nfoil
// bound vortices staged behind the wake at index n (LUDVM.py:1106, :1115, :1124: <= 256 sources, so the
// sum is done right here, one target per thread, instead of in a launch of its own): float64 update of
// the master copy, refresh of the fp32 mirrors (LUDVM.py:1108-1127).  HILO as in the pair kernels.
template <bool HILO>
__global__ void __launch_bounds__(kBlock)
finish_wake_advect_sym(const float* acc_u, const float* acc_w, long long nt, int nfoil, float vc4, double dt, double* x64,
                       double* z64, float* xh, float* xl, float* zh, float* zl, const float* g32, double* u_out,
                       double* w_out, const long long* n_dev = nullptr, TailDuty td = TailDuty{}) {
  __shared__ float fx[kBlock], fz[kBlock], fg[kBlock], fxl[HILO ? kBlock : 1], fzl[HILO ? kBlock : 1];
  const int tid = threadIdx.x;
  if (n_dev) nt = *n_dev;          // device-resident march: the wake size lives on the device
  if (tid < nfoil) {
    fx[tid] = xh[nt + tid]; fz[tid] = zh[nt + tid]; fg[tid] = g32[nt + tid];
    if (HILO) { fxl[tid] = xl[nt + tid]; fzl[tid] = zl[nt + tid]; }
  }
  __syncthreads();
  tail_duty_block0(td, nt);
  const long long i = (long long)blockIdx.x * kBlock + tid;
  if (i >= nt) return;
  const float s = (float)kInv2PiD;
  float fu = 0.0f, fw = 0.0f;
  const float xi = xh[i], zi = zh[i];
  const float xil = HILO ? xl[i] : 0.0f, zil = HILO ? zl[i] : 0.0f;
  for (int j = 0; j < nfoil; ++j) {
    float dx = xi - fx[j], dz = zi - fz[j];
    if (HILO) { dx += xil - fxl[j]; dz += zil - fzl[j]; }
    const float r2 = __builtin_fmaf(dz, dz, dx * dx);
    const float k = fg[j] * __builtin_amdgcn_rsqf(__builtin_fmaf(r2, r2, vc4));
    fu = __builtin_fmaf(dz, k, fu);
    fw = __builtin_fmaf(dx, k, fw);
  }
  const float su = (acc_u[i] + fu) * s, sw = -(acc_w[i] + fw) * s;
  if (u_out) { u_out[i] = (double)su; w_out[i] = (double)sw; }
  const double xn = x64[i] + dt * (double)su;
  const double zn = z64[i] + dt * (double)sw;
  // all reads of the old mirrors by this block happened above (own entry only); other blocks read the foil
  // entries [nt, nt + nfoil), which are not written here
  x64[i] = xn;
  z64[i] = zn;
  split_hilo(xn, xh[i], xl[i]);
  split_hilo(zn, zh[i], zl[i]);
  tail_duty(td, i, nt, xn, zn);
}

}  // namespace ludvm
