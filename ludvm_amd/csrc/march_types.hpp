// Types of the device-resident time march shared by the host code (ctx.hpp: the context holds a MarchSetup) and its kernels
// (march_kernels.hpp, which march.hip alone includes).
#pragma once
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"

namespace ludvm {

constexpr int kMarchMaxPan = 256;    // chord points (one thread each in march_solve)
constexpr int kMarchMaxCoef = 64;

struct MarchState {
  long long n;          // wake vortices (FREE + TEV + LEV, shedding order)
  long long n_old[2];   // ... before a step's solve (= after the previous roll-up); step s reads slot s & 1 and
                        // its finisher writes slot (s + 1) & 1 (blocks of one launch read and write it)
  long long itev, ilev; // TEV / LEV shed so far
  int shed;             // the step just solved shed a LEV (LEV_shed[i] != -1)
  int tail;             // vortices appended by the step just solved (1 or 2; 0 before the first one)
  double lesp_crit, sum_tev, sum_lev;
  double place[4];      // coming step: tev_x, lev_x, tev_z, lev_z
  double pvel[6];       // what the wake induces there and at the origin: u_tev, u_lev, u_org, w_tev, w_lev, w_org
  double newv[6];       // vortices shed by the step just solved, before their roll-up: x0, x1, z0, z1, g0, g1
  double newvel[4];     // ... and their velocities u0, u1, w0, w1 (wake + each other + bound vortices)
  double sum_abs_g;     // sum |Gamma| over the wake: bounds the symmetric kernel's raw sums (SymScale)
  SymScale sc[2];       // fixed-point scale by step parity: march_solve of step s leaves sc[(s + 1) & 1] for the wake
                        // as it stands after that step's shedding; an overlapped step s (old wake x old wake beside
                        // its own solve) reads sc[s & 1], a serial symmetric step s reads sc[(s + 1) & 1]
  long long sym_bad;    // a symmetric launch met a non-finite partial sum (sticky: NaN from there on)
  double prevA[kMarchMaxCoef];
  double chord[6 * kMarchMaxPan];          // coming step: u1 | w1 | u_tev | w_tev | u_lev | w_lev at the chord points
  double tgt[2 * (kMarchMaxPan + 3)];      // targets of the chord launch: x[npan + 3] | z[npan + 3]
};

// Read-only description of a run, passed by value.
struct MarchSetup {
  int npan, ncoef;
  double U, chord, rho, dt, piv, kelvin0;   // kelvin0 = sum(Gamma_free) - IC  (:758)
  double vc4;
  int method;                               // 0: 'Faure' (closed forms), 1: 'Ramesh' (Newton iterations, :683-739, :807-914)
  int maxiter;
  double maxerror, epsilon;
  // packed tables (device): see ludvm_march_setup in include/ludvm_hip.h
  const double* detadx; const double* eta; const double* xpan; const double* cm1; const double* wq;
  const double* opcs; const double* hcsd; const double* wx; const double* cproj; const double* ssin;
};

// Host-mapped progress ring: march_solve of step s stores (s << 32 | wake size after s) in slot s % kProgressRing.
// The host reads the slot of a step it KNOWS to be finished (an event recorded behind it has completed), so the bound
// it derives for the launches it enqueues next depends on the call's arguments only, not on how far the host happens
// to run ahead: the launch geometry -- and with it every bit of the results -- repeats from run to run.
constexpr int kProgressRing = 1024;
constexpr int kMarchRowHead = 12;   // g_tev, g_lev, shed, bound, LESP_prev, LESP, Fn, Fs, M, slot, phantom u, w

inline TailDuty make_tail_duty(MarchState* S, long long step, const double* kin_next, int npan) {
  TailDuty td;
  td.place = S->place; td.tgt = S->tgt; td.n_old = &S->n_old[(step + 1) & 1]; td.tail = &S->tail; td.shed = &S->shed;
  td.kin_next = kin_next; td.npan = npan;
  td.hist_row = nullptr; td.hist_nmax = 0;
  return td;
}

}  // namespace ludvm
