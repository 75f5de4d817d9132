// Device-resident time march (LUDVM.time_loop, reference LUDVM.py:597-1171, 'Faure' method): the per-step
// solve that the per-step path does on the host runs here, so consecutive time steps can be enqueued without
// a host round trip.  All of it is float64 and small (Npanels = 80 chord points, Ncoeffs = 30): one
// workgroup per step.
//
// Per time step i the device runs, in this order,
//   pair_f64_few        fp64 partial sums of the wake at Npanels + 3 targets: the chord points of step i
//                       (:746, :921), the two points where step i will shed its TEV and candidate LEV, and the
//                       origin (where the reference keeps a zero-strength LEV slot that it convects, :1112-1118)
//   march_chord_finish  sums the partials; unit influences of those two vortices at the chord points
//                       (:751, :926-931)                                              -> MarchState
//   march_solve         T1/T2/T3 downwash rows, Gamma_TEV (and Gamma_LEV when |A0| reaches LESPcrit), Fourier
//                       coefficients, bound vorticity, loads; appends the shed vortices and stages the bound
//                       vortices behind the wake                                     (:743-1090)
//   roll-up             (:1095-1127) the pair kernels of pair_kernels.hpp / pair_sym_kernels.hpp with the wake
//                       size taken from MarchState; their Euler finisher also places the TEV / LEV of step i + 1
//                       (:680-681, :797-800) and stages its chord points (TailDuty)
// Round 6 (profiles/r06_march_chain_ab.txt): the waves of the first three raise their issue priority (they run beside a roll-up
// kernel that keeps every SIMD busy) and march_solve's ~14 dependent workgroup reductions are done in 2 rounds (3 with
// 'Ramesh'), each value by the same tree: the same bits, config 2 -0.07 s.  Measured and dropped: summing the slabs inside
// march_solve with 12 helper wavefronts (a 1024-thread workgroup has to find a whole CU free: config 2 +1.2 s) and
// 64-source tiles for the chord sums (more slabs than the shorter walks save).
// Once the wake is large enough for the symmetric kernel the first three run on a second stream BESIDE the bulk of
// the roll-up: old wake on old wake depends only on the positions after the previous roll-up, not on this step's
// solve.  The shed vortices are then handled apart: what the old wake induces on them comes from the two extra
// targets of the chord launch, what they induce on the old wake is added by march_finish_sym.
#pragma once
#include <hip/hip_runtime.h>
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"
#include "march_types.hpp"

namespace ludvm {

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

// K sums at once, each by block_sum's tree (the same bits per value): one pair of barriers for all of them instead of one
// pair each -- march_solve is a chain of reductions, and two barriers + six dependent shuffles per value were most of its
// time.  `scratch` holds 4 K doubles.
template <int K>
__device__ __forceinline__ void block_sum_n(double (&v)[K], double* scratch) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += __shfl_down(v[k], off, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) scratch[4 * k + (threadIdx.x >> 6)] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = (scratch[4 * k] + scratch[4 * k + 1]) + (scratch[4 * k + 2] + scratch[4 * k + 3]);
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

// Vatistas pair: velocity at (xp, zp) per unit circulation of a vortex at (xs, zs)
__device__ __forceinline__ void unit_pair_f64(double xp, double zp, double xs, double zs, double vc4, double& u, double& w) {
  const double dx = xp - xs, dz = zp - zs;
  const double r2 = __builtin_fma(dz, dz, dx * dx);
  const double s = kInv2PiD * rsqrt_f64(__builtin_fma(r2, r2, vc4));
  u = dz * s;
  w = -dx * s;
}

// 'Ramesh' residuals.  The downwash is linear in the circulations being solved for, W = T1 + gt T2 + gl T3, so the
// reference's residual  U c pi (A0 + A1 / 2) + kelvin + (gt + gl)  with A0, A1 projected from W (:692-700, :820-835)
// needs only the six projections p0[k] = cproj[0] . T_k / U, p1[k] = cproj[1] . T_k / U.
struct RameshProj { double p0[3], p1[3], ucpi, kelvin; };
__device__ __forceinline__ double ramesh_res(const RameshProj& r, double gt, double gl, double& A0) {
  A0 = r.p0[0] + gt * r.p0[1] + gl * r.p0[2];
  const double A1 = r.p1[0] + gt * r.p1[1] + gl * r.p1[2];
  return r.ucpi * (A0 + A1 / 2) + r.kelvin + (gt + gl);
}
// Newton with a forward-difference slope, start -1, as the reference iterates it (:683-739)
__device__ __forceinline__ double ramesh_tev(const RameshProj& r, double maxerror, int maxiter, double eps) {
  double f = 1.0, g = -1.0, A0;
  int niter = 1;
  while (fabs(f) > maxerror && niter < maxiter) {
    f = ramesh_res(r, g, 0.0, A0);
    const double fd = ramesh_res(r, g + eps, 0.0, A0);
    g = g - f / ((fd - f) / eps);
    niter += 1;
  }
  return g;
}
// 2x2 Newton for (Gamma_LEV, Gamma_TEV) with the LESP condition as second equation (:807-914)
__device__ __forceinline__ void ramesh_tev_lev(const RameshProj& r, double lesp_crit, double guess, double maxerror, int maxiter,
                                               double eps, double& g_tev, double& g_lev) {
  g_tev = guess; g_lev = guess;
  double f1 = 0.1, f2 = 0.1;
  int niter = 1;
  while ((fabs(f1) > maxerror || fabs(f2) > maxerror) && niter < maxiter) {
    double A0;
    f1 = ramesh_res(r, g_tev, g_lev, A0);
    f2 = lesp_crit - A0;
    const double f1t = ramesh_res(r, g_tev + eps, g_lev, A0);
    const double f2t = lesp_crit - A0;
    const double f1l = ramesh_res(r, g_tev, g_lev + eps, A0);
    const double f2l = lesp_crit - A0;
    double dl, dt_;
    solve2((f1l - f1) / eps, (f1t - f1) / eps, (f2l - f2) / eps, (f2t - f2) / eps, f1, f2, dl, dt_);
    g_lev -= dl;
    g_tev -= dt_;
    niter += 1;
  }
}

// Start of a march call: the caller supplied the placements of the first step; stage its targets, and sum |Gamma|
// over the wake as it stands (fixed order) for the symmetric kernel's fixed-point scale.
__global__ void __launch_bounds__(kBlock)
march_begin(MarchState* S, const double* kin, int npan, int slot, const double* g64, double vc4) {
  __shared__ double part[kBlock / 64];
  const int t = threadIdx.x;
  const int ntt = npan + 3;
  const long long n = S->n;
  double a = 0.0;
  for (long long i = t; i < n; i += kBlock) a += fabs(g64[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if ((t & 63) == 0) part[t >> 6] = a;
  __syncthreads();
  if (t < npan) { S->tgt[t] = kin[7 + t]; S->tgt[ntt + t] = kin[7 + npan + t]; }
  if (t == 0) {
    S->tgt[npan] = S->place[0]; S->tgt[npan + 1] = S->place[1]; S->tgt[npan + 2] = 0.0;
    S->tgt[ntt + npan] = S->place[2]; S->tgt[ntt + npan + 1] = S->place[3]; S->tgt[ntt + npan + 2] = 0.0;
    S->n_old[slot] = n;
    const double tot = (part[0] + part[1]) + (part[2] + part[3]);
    S->sum_abs_g = tot;
    S->sym_bad = 0;
    sym_scale_from_sum(tot, vc4, &S->sc[slot], &S->sym_bad);
    S->sc[slot ^ 1] = S->sc[slot];
  }
}

// Sums the fp64 partial slabs of the chord launch (one WAVEFRONT per output column, fixed shuffle tree) and
// evaluates the unit influences of the coming TEV / candidate LEV at the chord points.  Columns: component k
// (0: u, 1: w) x target p; p < npan is a chord point, p = npan, npan + 1 the two placements, npan + 2 the origin.
__global__ void __launch_bounds__(kBlock)
march_chord_finish(const double* part, long long nt_pad, int nsplit, const double* direct_u, int npan, MarchState* S,
                   double vc4) {
  __builtin_amdgcn_s_setprio(3);                 // (see march_solve)
  const long long gtid = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long col = gtid >> 6;
  const int lane = threadIdx.x & 63;
  const int ntt = npan + 3;
  if (col >= 2 * ntt) return;   // whole wavefronts leave together
  const int k = (int)(col / ntt), p = (int)(col - (long long)k * ntt);
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
  if (p >= npan) {
    S->pvel[k * 3 + (p - npan)] = acc;
    return;
  }
  double* out = S->chord;
  out[k * npan + p] = acc;
  // unit vortex k (0: TEV, 1: LEV candidate) at chord point p
  double uu, ww;
  unit_pair_f64(S->tgt[p], S->tgt[ntt + p], S->place[k], S->place[2 + k], vc4, uu, ww);
  out[2 * npan + (k * 2 + 0) * npan + p] = uu;
  out[2 * npan + (k * 2 + 1) * npan + p] = ww;
}

// One workgroup.  kin = this step's kinematics row [alpha, alpha_dot, h_dot, te_x, te_z, le_x, le_z,
// xg[npan], zg[npan]]; row = this step's output row (kMarchRowHead + 2 ncoef + 2 npan doubles);
// progress (host-mapped, may be null) receives (step << 32 | n) so the host can bound the wake size of the
// steps it enqueues next without synchronizing.
__global__ void __launch_bounds__(kBlock)
march_solve(MarchSetup m, MarchState* S, const double* kin, double* row, long long step, double* x64, double* z64,
            double* g64, Mirrors mir, float* g32, unsigned long long* progress) {
  __shared__ double Wn[kMarchMaxPan];
  __shared__ double A[kMarchMaxCoef], Ad[kMarchMaxCoef];
  __shared__ double scratch[4 * 8];
  // the chain runs beside a roll-up kernel that keeps every SIMD's issue slots busy: its few waves go first
  __builtin_amdgcn_s_setprio(3);
  const int j = threadIdx.x;
  const int npan = m.npan, ncoef = m.ncoef;
  const bool on = j < npan;
  const double pi = 3.14159265358979323846;
  const double al = kin[0], ald = kin[1], hd = kin[2];
  const double ca = cos(al), sa = sin(al);
  const double* xg = kin + 7;
  const double* zg = kin + 7 + npan;

  // what the solve reads of the state, before anything is rewritten
  const long long n0 = S->n;
  const double tev_x = S->place[0], lev_x = S->place[1], tev_z = S->place[2], lev_z = S->place[3];
  const double pu0 = S->pvel[0], pu1 = S->pvel[1], pw0 = S->pvel[3], pw1 = S->pvel[4];
  const double puo = S->pvel[2], pwo = S->pvel[5];

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
  // the six chord integrals at once (I3 and the J's are needed on shedding steps only: summed anyway, one pair of barriers)
  double ij[6] = {t1 * cm1, t2 * cm1, t3 * cm1, t1 * wq, t2 * wq, t3 * wq};
  block_sum_n<6>(ij, scratch);
  const double I1 = ij[0], I2 = ij[1];
  const double kelvin = S->sum_tev + S->sum_lev + m.kelvin0;
  const bool ramesh = m.method == 1;
  RameshProj rp{};
  double g_tev;
  if (ramesh) {
    const double c0 = on ? m.cproj[j] / m.U : 0.0, c1 = on ? m.cproj[npan + j] / m.U : 0.0;
    double pr[6] = {t1 * c0, t2 * c0, t3 * c0, t1 * c1, t2 * c1, t3 * c1};
    block_sum_n<6>(pr, scratch);
    rp.p0[0] = pr[0]; rp.p0[1] = pr[1]; rp.p0[2] = pr[2];
    rp.p1[0] = pr[3]; rp.p1[1] = pr[4]; rp.p1[2] = pr[5];
    rp.ucpi = m.U * m.chord * pi;
    rp.kelvin = kelvin;
    g_tev = ramesh_tev(rp, m.maxerror, m.maxiter, m.epsilon);
  } else {
    g_tev = -(I1 + kelvin) / (1 + I2);               // :758-760
  }
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
  const double ucpi = m.U * m.chord * pi;
  double bound = ramesh ? ucpi * (A[0] + A[1] / 2) : I1 + g_tev * I2;     // :761 / :738
  const double lesp_prev = A[0];
  double lesp_crit = S->lesp_crit;
  const bool shed = fabs(A[0]) >= fabs(lesp_crit);   // :781
  __syncthreads();                                   // everyone has read A[0] before it is rewritten
  if (shed) {
    lesp_crit = A[0] < 0 ? -fabs(lesp_crit) : fabs(lesp_crit);     // :802-805
    const double I3 = ij[2];
    const double J1 = -1 / pi * ij[3];
    const double J2 = -1 / pi * ij[4];
    const double J3 = -1 / pi * ij[5];
    if (ramesh) ramesh_tev_lev(rp, lesp_crit, g_tev, m.maxerror, m.maxiter, m.epsilon, g_tev, g_lev);
    else solve2(1 + I2, 1 + I3, J2, J3, -(I1 + kelvin), lesp_crit - J1, g_tev, g_lev);   // :944-954
    if (on) Wn[j] = (t1 + g_tev * t2 + g_lev * t3) / m.U;
    __syncthreads();
    if (j < ncoef) {
      double acc = 0.0;
      const double* cp = m.cproj + (long long)j * npan;
      for (int q = 0; q < npan; ++q) acc = __builtin_fma(cp[q], Wn[q], acc);
      // 'Faure' takes A0 from the LESP form (:959); derivatives keep their values in both methods (:963-966)
      A[j] = (j == 0 && !ramesh) ? J1 + g_tev * J2 + g_lev * J3 : acc;
    }
    __syncthreads();
    bound = ramesh ? ucpi * (A[0] + A[1] / 2) : I1 + g_tev * I2 + g_lev * I3;
  }

  // bound vorticity per panel (:987-1010)
  double gamma = 0.0, dgamma = 0.0;
  if (on) {
    double ssum = 0.0;
    for (int q = 1; q < ncoef; ++q) ssum = __builtin_fma(A[q], m.ssin[(long long)(q - 1) * npan + j], ssum);
    gamma = 2 * m.U * (A[0] * m.opcs[j] + ssum);
    dgamma = gamma * m.hcsd[j];
  }
  const int k = shed ? 2 : 1;
  // One more round of sums: the loads (:1035-1090; tangential velocity on the chord from the whole wake by linearity), and
  // the velocity of the vortices shed now, for the roll-up that treats them apart (:1105-1124 restricted to them): the
  // wake's part came with the chord sums; the bound vortices' part is summed here; plus each other.
  // The reference convects LEV slot `ilev` -- zero strength, at the origin -- on a step that sheds no LEV and stores
  // where it lands in path['LEV'][i] (:1112-1118); its velocity: wake (chord launch), the new TEV, the bound vortices.
  double r8[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // fn, m, fu0, fw0, fu1, fw1, fuo, fwo
  if (on) {
    const double uc1 = u1 + g_tev * ut1 + (shed ? g_lev * ul1 : 0.0);
    const double wc1 = w1 + g_tev * wt1 + (shed ? g_lev * wl1 : 0.0);
    const double u = uc1 * ca - wc1 * sa;
    r8[0] = u * gamma * m.wx[j];
    r8[1] = u * gamma * m.xpan[j] * m.wx[j];
    double uu, ww;
    unit_pair_f64(tev_x, tev_z, xg[j], zg[j], m.vc4, uu, ww);
    r8[2] = dgamma * uu; r8[3] = dgamma * ww;
    if (shed) {
      unit_pair_f64(lev_x, lev_z, xg[j], zg[j], m.vc4, uu, ww);
      r8[4] = dgamma * uu; r8[5] = dgamma * ww;
    } else {
      unit_pair_f64(0.0, 0.0, xg[j], zg[j], m.vc4, uu, ww);
      r8[6] = dgamma * uu; r8[7] = dgamma * ww;
    }
  }
  block_sum_n<8>(r8, scratch);
  const double fn_sum = r8[0], m_sum = r8[1];
  const double su0 = r8[2], sw0 = r8[3], su1 = r8[4], sw1 = r8[5], suo = r8[6], swo = r8[7];
  __syncthreads();                                   // all reads of S are done; it is rewritten below

  // Local-origin mirrors of the entries written now -- wake index n0 (TEV), n0 + 1 (LEV when shed), then the npan
  // bound vortices: an origin class (256-vortex block x index parity) whose first member is among them takes its origin
  // from the entry written there, the others keep the origin the last Euler finisher gave them.
  auto origin_of = [&](long long i, float& ox, float& oz) {
    const long long b = i >> kOriginShift, cs = (b << kOriginShift) + (i & 1), slot = origin_slot(i);
    if (cs >= n0) {
      double bx, bz;
      if (cs == n0) { bx = tev_x; bz = tev_z; }
      else if (shed && cs == n0 + 1) { bx = lev_x; bz = lev_z; }
      else { bx = xg[cs - n0 - k]; bz = zg[cs - n0 - k]; }
      ox = (float)bx; oz = (float)bz;
      if (i == cs) { mir.cx[slot] = ox; mir.cz[slot] = oz; }
      // a block opened by the very last entry written now: its still empty odd class gets a number too
      if (i == cs && (i & 1) == 0 && i == n0 + k + npan - 1) { mir.cx[slot + 1] = ox; mir.cz[slot + 1] = oz; }
    } else {
      ox = mir.cx[slot]; oz = mir.cz[slot];
    }
  };

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
    row[10] = 0.0; row[11] = 0.0;
    if (!shed) {
      double uu, ww;
      unit_pair_f64(0.0, 0.0, tev_x, tev_z, m.vc4, uu, ww);
      row[10] = puo + suo + g_tev * uu;
      row[11] = pwo + swo + g_tev * ww;
    }
    // the shed vortices join the wake (:1095-1098)
    float ox, oz;
    origin_of(n0, ox, oz);
    x64[n0] = tev_x; z64[n0] = tev_z; g64[n0] = g_tev;
    store_mirrors(mir, n0, tev_x, tev_z, ox, oz); g32[n0] = (float)g_tev;
    double m01u = 0, m01w = 0, m10u = 0, m10w = 0;   // TEV <- LEV, LEV <- TEV
    if (shed) {
      const long long n1 = n0 + 1;
      origin_of(n1, ox, oz);
      x64[n1] = lev_x; z64[n1] = lev_z; g64[n1] = g_lev;
      store_mirrors(mir, n1, lev_x, lev_z, ox, oz); g32[n1] = (float)g_lev;
      double uu, ww;
      unit_pair_f64(tev_x, tev_z, lev_x, lev_z, m.vc4, uu, ww);
      m01u = g_lev * uu; m01w = g_lev * ww;
      unit_pair_f64(lev_x, lev_z, tev_x, tev_z, m.vc4, uu, ww);
      m10u = g_tev * uu; m10w = g_tev * ww;
    }
    S->newv[0] = tev_x; S->newv[1] = lev_x; S->newv[2] = tev_z; S->newv[3] = lev_z;
    S->newv[4] = g_tev; S->newv[5] = shed ? g_lev : 0.0;
    S->newvel[0] = pu0 + su0 + m01u; S->newvel[2] = pw0 + sw0 + m01w;
    S->newvel[1] = pu1 + su1 + m10u; S->newvel[3] = pw1 + sw1 + m10w;
    S->n = n0 + k;
    S->itev += 1;
    S->ilev += shed ? 1 : 0;
    S->shed = shed ? 1 : 0;
    S->tail = k;
    S->lesp_crit = lesp_crit;
    S->sum_tev += g_tev;
    S->sum_lev += g_lev;
    const double sabs = S->sum_abs_g + fabs(g_tev) + fabs(g_lev);
    S->sum_abs_g = sabs;
    sym_scale_from_sum(sabs, m.vc4, &S->sc[(step + 1) & 1], &S->sym_bad);
    if (progress) {
      __atomic_store_n(progress + (step % kProgressRing), ((unsigned long long)step << 32) | (unsigned long long)(n0 + k),
                       __ATOMIC_RELAXED);
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
    float ox, oz;
    origin_of(i, ox, oz);
    x64[i] = x; z64[i] = z; g64[i] = dgamma;
    store_mirrors(mir, i, x, z, ox, oz); g32[i] = (float)dgamma;
  }
}

// Euler finisher of the overlapped roll-up.  The symmetric kernel ran on the wake as it was BEFORE this step's
// solve ([0, n_old), raw fixed-point sums in acc_u / acc_w, scale sc); here every old vortex also feels the vortices
// shed this step (S->newv) and the bound vortices (staged at [n, n + nfoil)), and the shed vortices move with the
// velocities march_solve left in S->newvel.  The raw sums are zeroed after use, so the accumulators need no memset
// between steps.  One workgroup = one origin block (see finish_wake_advect).
__global__ void __launch_bounds__(kFinBlock)
march_finish_sym(long long* acc_u, long long* acc_w, const SymScale* sc, MarchState* S, const long long* n_old_p,
                 int nfoil, float vc4, double dt, double* x64, double* z64, Mirrors m, const float* g32, TailDuty td,
                 long long* bad_step = nullptr, long long* bad_next = nullptr) {
  // bad_step / bad_next (sharded roll-up): this step's count of non-finite partial sums, summed over all owners by
  // the all-reduce that also summed acc_u / acc_w, and the next step's counter, cleared here; S->sym_bad keeps it
  __shared__ float org[4];
  const bool bad = S->sym_bad != 0 || (bad_step && *bad_step != 0);
  const long long n = S->n, n_old = *n_old_p;
  const int k = (int)(n - n_old);
  tail_duty_block0(td, n);
  const long long i = (long long)blockIdx.x * kFinBlock + threadIdx.x;
  const bool on = i < n, old = i < n_old;
  const double xo = on ? x64[i] : 0.0, zo = on ? z64[i] : 0.0;
  float fu, fw;
  staged_sources_on(old, xo, zo, x64, z64, g32, n, nfoil, vc4, fu, fw, k, S->newv);
  double xn = 0.0, zn = 0.0;
  if (on) {
    double su, sw;
    if (old) {
      const float s = (float)kInv2PiD;
      su = (double)((fx_read(acc_u, i, sc, bad) + fu) * s);
      sw = (double)(-(fx_read(acc_w, i, sc, bad) + fw) * s);
      acc_u[i] = 0;
      acc_w[i] = 0;
    } else {
      const int q = (int)(i - n_old);
      su = S->newvel[q];
      sw = S->newvel[2 + q];
    }
    xn = xo + dt * su;
    zn = zo + dt * sw;
    publish_origins(m, i, n, xn, zn, org);
  }
  __syncthreads();
  if (bad_step && blockIdx.x == 0 && threadIdx.x == 0) {
    if (bad) S->sym_bad = 1;       // (every block has read S->sym_bad OR *bad_step: both say the same from here on)
    *bad_next = 0;
  }
  if (!on) return;
  x64[i] = xn;
  z64[i] = zn;
  store_mirrors(m, i, xn, zn, org[i & 1], org[2 + (i & 1)]);
  tail_duty(td, i, n, xn, zn);
}

}  // namespace ludvm
