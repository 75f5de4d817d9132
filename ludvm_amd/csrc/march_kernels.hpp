// Device-resident time march (LUDVM.time_loop, reference LUDVM.py:597-1171, 'Faure' method): the per-step
// solve that the per-step path does on the host runs here, so consecutive time steps can be enqueued without
// a host round trip.  All of it is float64 and small (Npanels = 80 chord points, Ncoeffs = 30): one
// workgroup per step.
//
// Per step i the device runs
//   march_solve         T1/T2/T3 downwash rows from the chord sums, Gamma_TEV (and Gamma_LEV when |A0| reaches
//                       LESPcrit), Fourier coefficients, bound vorticity, loads; appends the shed vortices and
//                       stages the bound vortices behind the wake                    (:743-1090)
//   roll-up             the pair kernels of pair_kernels.hpp / pair_sym_kernels.hpp with the wake size taken
//                       from MarchState (n_dev)                                      (:1095-1127)
//   pair_f64<128>       fp64 wake -> chord partial sums for step i+1                 (:746, :921)
//   march_chord_finish  sums the partials, places the next TEV / candidate LEV, unit influences (:672-681,
//                       :751, :788-800, :926-931), all into MarchState
#pragma once
#include <hip/hip_runtime.h>
#include "pair_kernels.hpp"

namespace ludvm {

constexpr int kMarchMaxPan = 256;    // chord points (one thread each in march_solve)
constexpr int kMarchMaxCoef = 64;

struct MarchState {
  long long n;          // wake vortices (FREE + TEV + LEV, shedding order)
  long long itev, ilev; // TEV / LEV shed so far
  int shed;             // the step just solved shed a LEV (LEV_shed[i] != -1)
  int tail;             // vortices appended by the step just solved (1 or 2; 0 before the first one)
  double lesp_crit, sum_tev, sum_lev;
  double place[4];      // coming step: tev_x, lev_x, tev_z, lev_z
  double prevA[kMarchMaxCoef];
  double chord[6 * kMarchMaxPan];   // coming step: u1 | w1 | u_tev | w_tev | u_lev | w_lev at the chord points
};

// Read-only description of a run, passed by value.
struct MarchSetup {
  int npan, ncoef;
  double U, chord, rho, dt, piv, kelvin0;   // kelvin0 = sum(Gamma_free) - IC  (:758)
  // packed tables (device): see ludvm_march_setup in include/ludvm_hip.h
  const double* detadx; const double* eta; const double* xpan; const double* cm1; const double* wq;
  const double* opcs; const double* hcsd; const double* wx; const double* cproj; const double* ssin;
};

constexpr int kMarchRowHead = 10;   // g_tev, g_lev, shed, bound, LESP_prev, LESP, Fn, Fs, M, slot

// Sum v over the workgroup (fixed tree: wavefront shuffles, then the 4 wave partials in order); every thread
// gets the result.  `scratch` holds kBlock / 64 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// 2x2 solve as LAPACK's dgesv does it (partial pivoting), so that the result has the rounding of
// np.linalg.solve in the reference (:944-954).
__device__ __forceinline__ void solve2(double a00, double a01, double a10, double a11, double b0, double b1, double& x0,
                                       double& x1) {
  if (fabs(a10) > fabs(a00)) {
    double t;
    t = a00; a00 = a10; a10 = t;
    t = a01; a01 = a11; a11 = t;
    t = b0; b0 = b1; b1 = t;
  }
  const double l = a10 / a00;
  const double u11 = a11 - l * a01;
  const double y1 = b1 - l * b0;
  x1 = y1 / u11;
  x0 = (b0 - a01 * x1) / a00;
}

// One workgroup.  kin = this step's kinematics row [alpha, alpha_dot, h_dot, te_x, te_z, le_x, le_z,
// xg[npan], zg[npan]]; row = this step's output row (kMarchRowHead + 2 ncoef + 2 npan doubles);
// progress (host-mapped, may be null) receives (step << 32 | n) so the host can bound the wake size of the
// steps it enqueues next without synchronizing.
__global__ void __launch_bounds__(kBlock)
march_solve(MarchSetup m, MarchState* S, const double* kin, double* row, long long step, double* x64, double* z64,
            double* g64, float* xh, float* xl, float* zh, float* zl, float* g32, unsigned long long* progress) {
  __shared__ double Wn[kMarchMaxPan];
  __shared__ double A[kMarchMaxCoef], Ad[kMarchMaxCoef];
  __shared__ double scratch[kBlock / 64];
  const int j = threadIdx.x;
  const int npan = m.npan, ncoef = m.ncoef;
  const bool on = j < npan;
  const double pi = 3.14159265358979323846;
  const double al = kin[0], ald = kin[1], hd = kin[2];
  const double ca = cos(al), sa = sin(al);
  const double* xg = kin + 7;
  const double* zg = kin + 7 + npan;

  double u1 = 0, w1 = 0, ut1 = 0, wt1 = 0, ul1 = 0, wl1 = 0, dedx = 0, cm1 = 0, wq = 0;
  if (on) {
    u1 = S->chord[j]; w1 = S->chord[npan + j];
    ut1 = S->chord[2 * npan + j]; wt1 = S->chord[3 * npan + j];
    ul1 = S->chord[4 * npan + j]; wl1 = S->chord[5 * npan + j];
    dedx = m.detadx[j]; cm1 = m.cm1[j]; wq = m.wq[j];
  }
  // chord frame (:586-587) and the downwash rows: T1 from the existing wake + kinematics (:588-593), T2 / T3
  // from the unit TEV / LEV (only the induced part enters)
  double t1 = 0, t2 = 0, t3 = 0;
  if (on) {
    const double u = u1 * ca - w1 * sa, w = u1 * sa + w1 * ca;
    t1 = dedx * (m.U * ca + hd * sa + u - ald * m.eta[j]) - m.U * sa - ald * (m.xpan[j] - m.piv) + hd * ca - w;
    const double ut = ut1 * ca - wt1 * sa, un = ut1 * sa + wt1 * ca;
    t2 = dedx * ut - un;
    const double ult = ul1 * ca - wl1 * sa, uln = ul1 * sa + wl1 * ca;
    t3 = dedx * ult - uln;
  }
  const double I1 = block_sum(t1 * cm1, scratch);
  const double I2 = block_sum(t2 * cm1, scratch);
  const double kelvin = S->sum_tev + S->sum_lev + m.kelvin0;
  double g_tev = -(I1 + kelvin) / (1 + I2);          // :758-760
  double g_lev = 0.0;
  if (on) Wn[j] = (t1 + g_tev * t2) / m.U;
  __syncthreads();
  if (j < ncoef) {
    double acc = 0.0;
    const double* cp = m.cproj + (long long)j * npan;
    for (int q = 0; q < npan; ++q) acc = __builtin_fma(cp[q], Wn[q], acc);
    A[j] = acc;
    Ad[j] = (acc - S->prevA[j]) / m.dt;              // :772-773
  }
  __syncthreads();
  double bound = I1 + g_tev * I2;
  const double lesp_prev = A[0];
  double lesp_crit = S->lesp_crit;
  const bool shed = fabs(A[0]) >= fabs(lesp_crit);   // :781
  __syncthreads();                                   // everyone has read A[0] before it is rewritten
  if (shed) {
    lesp_crit = A[0] < 0 ? -fabs(lesp_crit) : fabs(lesp_crit);     // :802-805
    const double I3 = block_sum(t3 * cm1, scratch);
    const double J1 = -1 / pi * block_sum(t1 * wq, scratch);
    const double J2 = -1 / pi * block_sum(t2 * wq, scratch);
    const double J3 = -1 / pi * block_sum(t3 * wq, scratch);
    solve2(1 + I2, 1 + I3, J2, J3, -(I1 + kelvin), lesp_crit - J1, g_tev, g_lev);   // :944-954
    if (on) Wn[j] = (t1 + g_tev * t2 + g_lev * t3) / m.U;
    __syncthreads();
    if (j < ncoef) {
      double acc = 0.0;
      const double* cp = m.cproj + (long long)j * npan;
      for (int q = 0; q < npan; ++q) acc = __builtin_fma(cp[q], Wn[q], acc);
      A[j] = j == 0 ? J1 + g_tev * J2 + g_lev * J3 : acc;          // :959; derivatives keep their values (:963-966)
    }
    bound = I1 + g_tev * I2 + g_lev * I3;
    __syncthreads();
  }

  // bound vorticity per panel (:987-1010)
  double gamma = 0.0, dgamma = 0.0;
  if (on) {
    double ssum = 0.0;
    for (int q = 1; q < ncoef; ++q) ssum = __builtin_fma(A[q], m.ssin[(long long)(q - 1) * npan + j], ssum);
    gamma = 2 * m.U * (A[0] * m.opcs[j] + ssum);
    dgamma = gamma * m.hcsd[j];
  }
  // loads (:1035-1090); tangential velocity on the chord from the whole wake by linearity
  double fn_t = 0.0, m_t = 0.0;
  if (on) {
    const double uc1 = u1 + g_tev * ut1 + (shed ? g_lev * ul1 : 0.0);
    const double wc1 = w1 + g_tev * wt1 + (shed ? g_lev * wl1 : 0.0);
    const double u = uc1 * ca - wc1 * sa;
    fn_t = u * gamma * m.wx[j];
    m_t = u * gamma * m.xpan[j] * m.wx[j];
  }
  const double fn_sum = block_sum(fn_t, scratch);
  const double m_sum = block_sum(m_t, scratch);

  const long long n0 = S->n;
  const int k = shed ? 2 : 1;
  const double tev_x = S->place[0], lev_x = S->place[1], tev_z = S->place[2], lev_z = S->place[3];
  __syncthreads();                                   // all reads of S are done; it is rewritten below

  if (j == 0) {
    const double c = m.chord, U = m.U, rho = m.rho;
    const double A0 = A[0], A1 = A[1], A2 = A[2];
    const double A0d = Ad[0], A1d = Ad[1], A2d = Ad[2], A3d = Ad[3];
    const double Ueff = U * ca + hd * sa;
    const double Fn = rho * pi * c * U * (Ueff * (A0 + 0.5 * A1) + c * (3.0 / 4 * A0d + 1.0 / 4 * A1d + 1.0 / 8 * A2d))
        + rho * fn_sum;
    const double Fs = rho * pi * c * U * U * A0 * A0;
    const double M = m.piv * Fn - rho * pi * c * c * U * (Ueff * (1.0 / 4 * A0 + 1.0 / 4 * A1 - 1.0 / 8 * A2)
        + c * (7.0 / 16 * A0d + 3.0 / 16 * A1d + 1.0 / 16 * A2d - 1.0 / 64 * A3d)) - rho * m_sum;
    row[0] = g_tev; row[1] = g_lev; row[2] = shed ? 1.0 : 0.0; row[3] = bound; row[4] = lesp_prev; row[5] = A0;
    row[6] = Fn; row[7] = Fs; row[8] = M; row[9] = (double)n0;
    // the shed vortices join the wake (:1095-1098)
    x64[n0] = tev_x; z64[n0] = tev_z; g64[n0] = g_tev;
    split_hilo(tev_x, xh[n0], xl[n0]); split_hilo(tev_z, zh[n0], zl[n0]); g32[n0] = (float)g_tev;
    if (shed) {
      const long long n1 = n0 + 1;
      x64[n1] = lev_x; z64[n1] = lev_z; g64[n1] = g_lev;
      split_hilo(lev_x, xh[n1], xl[n1]); split_hilo(lev_z, zh[n1], zl[n1]); g32[n1] = (float)g_lev;
    }
    S->n = n0 + k;
    S->itev += 1;
    S->ilev += shed ? 1 : 0;
    S->shed = shed ? 1 : 0;
    S->tail = k;
    S->lesp_crit = lesp_crit;
    S->sum_tev += g_tev;
    S->sum_lev += g_lev;
    if (progress) {
      __atomic_store_n(progress, ((unsigned long long)step << 32) | (unsigned long long)(n0 + k), __ATOMIC_RELAXED);
    }
  }
  if (j < ncoef) {
    S->prevA[j] = A[j];
    row[kMarchRowHead + j] = A[j];
    row[kMarchRowHead + ncoef + j] = Ad[j];
  }
  if (on) {
    row[kMarchRowHead + 2 * ncoef + j] = gamma;
    row[kMarchRowHead + 2 * ncoef + npan + j] = dgamma;
    // bound vortices ride behind the wake as sources of the roll-up (:1106, :1115, :1124)
    const long long i = n0 + k + j;
    const double x = xg[j], z = zg[j];
    x64[i] = x; z64[i] = z; g64[i] = dgamma;
    split_hilo(x, xh[i], xl[i]); split_hilo(z, zh[i], zl[i]); g32[i] = (float)dgamma;
  }
}

// chord_finish_f64 (pair_kernels.hpp) for the march: wake size, tail length and the LEV flag come from
// MarchState, and the results (chord sums, unit influences, placements of the coming step) go back into it.
// geo = [te_x, te_z, le_x, le_z] of the coming step; xt / zt its chord points.  tail == 0 (start of a
// march): the placements already in S are used as they are.
__global__ void __launch_bounds__(kBlock)
march_chord_finish(const double* part, long long nt_pad, int nsplit, const double* direct_u, const double* xt,
                   const double* zt, long long nt, const double* x64, const double* z64, MarchState* S, const double* geo,
                   double vc4) {
  const long long gtid = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long col = gtid >> 6;
  const int lane = threadIdx.x & 63;
  const long long n = S->n;
  const int tail = S->tail;
  double ux[2], uz[2];
  if (tail == 0) {
    ux[0] = S->place[0]; ux[1] = S->place[1]; uz[0] = S->place[2]; uz[1] = S->place[3];
  } else {
    const double tex = geo[0], tez = geo[1], lex = geo[2], lez = geo[3];
    const long long it = n - tail;
    ux[0] = tex + (x64[it] - tex) / 3;
    uz[0] = tez + (z64[it] - tez) / 3;
    if (S->shed && tail == 2) {
      ux[1] = lex + (x64[n - 1] - lex) / 3;
      uz[1] = lez + (z64[n - 1] - lez) / 3;
    } else {
      ux[1] = lex;
      uz[1] = lez;
    }
  }
  // no hazard on S->place: it is read above only when tail == 0 and written here only when tail != 0
  if (gtid == 0 && tail != 0) { S->place[0] = ux[0]; S->place[1] = ux[1]; S->place[2] = uz[0]; S->place[3] = uz[1]; }
  if (col >= 2 * nt) return;   // whole wavefronts leave together
  const long long k = col / nt, p = col - k * nt;
  double acc = 0.0;
  if (part != nullptr) {
    const double* c0 = part + k * nt_pad + p;
    for (int sidx = lane; sidx < nsplit; sidx += 64) acc += c0[(long long)sidx * 2 * nt_pad];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  } else if (nsplit == 1) {
    acc = direct_u[k * nt_pad + p];
  }
  if (lane != 0) return;
  double* out = S->chord;
  out[k * nt + p] = acc;
  const double dx = xt[p] - ux[k];
  const double dz = zt[p] - uz[k];
  const double r2 = __builtin_fma(dz, dz, dx * dx);
  const double s = kInv2PiD * rsqrt_f64(__builtin_fma(r2, r2, vc4));
  out[2 * nt + (k * 2 + 0) * nt + p] = dz * s;
  out[2 * nt + (k * 2 + 1) * nt + p] = -dx * s;
}

}  // namespace ludvm
