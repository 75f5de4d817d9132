// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: the device-resident time march (LUDVM.time_loop, LUDVM.py:597-1171, with the solve on the device).
#include "ctx.hpp"
#include "march_kernels.hpp"

extern "C" {

/* ---- device-resident time march ------------------------------------------------------------- */

int ludvm_march_setup(ludvm_ctx* c, int npan, int ncoef, const double* scalars, const double* tables, const double* kin,
                      size_t kin_rows) {
  if (!c) return LUDVM_E_ARG;
  c->march_ready = false;
  if (!scalars || !tables || !kin) return fail(c, LUDVM_E_ARG, "null array");
  if (npan < 1 || npan > kMarchMaxPan || ncoef < 4 || ncoef > kMarchMaxCoef)
    return fail(c, LUDVM_E_ARG, "march: 1 <= Npanels <= 256 and 4 <= Ncoeffs <= 64");
  if (kin_rows < 2) return fail(c, LUDVM_E_ARG, "march: kinematics table too short");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t P = (size_t)npan;
  const size_t tab_doubles = 8 * P + (size_t)ncoef * P + (size_t)(ncoef - 1) * P;
  const size_t kin_doubles = kin_rows * (7 + 2 * P);
  CHK(ensure(c, c->march_tab, tab_doubles * 8));
  CHK(ensure(c, c->march_kin, kin_doubles * 8));
  CHK(ensure(c, c->march_state, sizeof(MarchState)));
  HIPCHK(c, hipMemcpyAsync(c->march_tab.p, tables, tab_doubles * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->march_kin.p, kin, kin_doubles * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  MarchSetup& m = c->msetup;
  m.npan = npan; m.ncoef = ncoef;
  m.U = scalars[0]; m.chord = scalars[1]; m.rho = scalars[2]; m.dt = scalars[3]; m.piv = scalars[4];
  c->march_vcore = scalars[5];
  m.kelvin0 = scalars[7] - scalars[6];          // sum(Gamma_free) - IC
  m.vc4 = (scalars[5] * scalars[5]) * (scalars[5] * scalars[5]);
  m.method = scalars[8] != 0.0 ? 1 : 0;
  m.maxerror = scalars[9]; m.maxiter = (int)scalars[10]; m.epsilon = scalars[11];
  if (m.method == 1 && !(m.maxerror > 0.0 && m.maxiter >= 1 && m.epsilon > 0.0))
    return fail(c, LUDVM_E_ARG, "march: 'Ramesh' needs maxerror > 0, maxiter >= 1, epsilon > 0");
  const double* t = static_cast<const double*>(c->march_tab.p);
  m.detadx = t; m.eta = t + P; m.xpan = t + 2 * P; m.cm1 = t + 3 * P; m.wq = t + 4 * P; m.opcs = t + 5 * P;
  m.hcsd = t + 6 * P; m.wx = t + 7 * P; m.cproj = t + 8 * P; m.ssin = t + 8 * P + (size_t)ncoef * P;
  c->march_kin_rows = kin_rows;
  if (!c->progress) {
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->progress), kProgressRing * sizeof(unsigned long long), hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->progress_dev), c->progress, 0));
  }
  for (auto& e : c->march_ev)
    if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  if (!c->ev_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  if (!c->stream_b) {
    int least = 0, greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    // the solve chain is short and latency-critical: let it cut in front of the roll-up's workgroups
    HIPCHK(c, hipStreamCreateWithPriority(&c->stream_b, hipStreamNonBlocking, greatest));
  }
  c->march_ready = true;
  return LUDVM_OK;
}

}  // extern "C"

namespace {

// fp64 partial sums of the wake at the npan + 3 targets staged in MarchState (chord points of the coming solve, its
// two placements, the origin), then the finisher that leaves sums and unit influences in the device state.  Launches on
// c->stream (the caller points it at the second stream for overlapped steps).
int march_chord_launch(ludvm_ctx* c, long long n_ub) {
  const MarchSetup& m = c->msetup;
  const size_t P = (size_t)m.npan, NT = P + 3;
  MarchState* S = static_cast<MarchState*>(c->march_state.p);
  PairArgs a{};
  a.xs = c->x64; a.zs = c->z64; a.gs = c->g64;
  a.ns = 0; a.n_dev = &S->n; a.ns_dev = 1; a.nt_dev = 0;
  a.xt = S->tgt; a.zt = S->tgt + NT; a.nt = (long long)NT;
  a.vc4 = m.vc4;
  Plan p = make_plan(c, (long long)NT, std::max<long long>(n_ub, 1), LUDVM_PREC_F64);
  const bool was = c->timing;
  c->timing = false;   // the chord sums are not the dominant kernel
  int rc = launch_pair(c, a, p, LUDVM_PREC_F64, nullptr, nullptr);
  c->timing = was;
  CHK(rc);
  const double* slab = static_cast<const double*>(c->part.p);
  hipLaunchKernelGGL(march_chord_finish, dim3(blocks_for((long long)(2 * NT * 64))), dim3(kBlock), 0, c->stream,
                     p.nsplit > 1 ? slab : (const double*)nullptr, p.nt_pad, p.nsplit, slab, (int)P, S, m.vc4);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

// workspace a march step may need when the wake holds at most n_ub vortices
void march_workspace(const ludvm_ctx* c, long long n_ub, int precision, size_t nfoil, size_t& part_bytes, size_t& acc_bytes) {
  const long long nt = std::max<long long>(n_ub, 1);
  // (both kernels' workspaces where either may run: the overlapped march switches to the symmetric kernel earlier
  // than the serial one)
  if (precision != LUDVM_PREC_F64 && use_symmetric(c, nt, c->msetup.vc4, true))
    acc_bytes = std::max(acc_bytes, (size_t)2 * (size_t)((nt + 63) / 64 * 64) * sizeof(long long));
  if (precision == LUDVM_PREC_F64 || !use_symmetric(c, nt, c->msetup.vc4, false)) {
    Plan p = make_plan(c, nt, nt + (long long)nfoil, precision);
    const size_t elt = precision == LUDVM_PREC_F64 ? 8 : 4;
    part_bytes = std::max(part_bytes, (size_t)p.nsplit * 2 * (size_t)p.nt_pad * elt);
  }
  Plan q = make_plan(c, (long long)nfoil + 3, nt, LUDVM_PREC_F64);
  part_bytes = std::max(part_bytes, (size_t)q.nsplit * 2 * (size_t)q.nt_pad * 8);
}

}  // namespace

extern "C" {

int ludvm_march_run(ludvm_ctx* c, long long first_step, long long count, int precision, double* state, double* rows,
                    double* hist, size_t hist_nmax, const long long* anchors) {
  if (!c) return LUDVM_E_ARG;
  if (!c->march_ready) return fail(c, LUDVM_E_STATE, "ludvm_march_setup has not been called");
  if (!valid_precision(precision)) return fail(c, LUDVM_E_ARG, "unknown precision");
  if (!state || !rows) return fail(c, LUDVM_E_ARG, "null array");
  if (count < 1 || first_step < 1 || (size_t)(first_step + count) > c->march_kin_rows)
    return fail(c, LUDVM_E_ARG, "march: steps outside the kinematics table");
  const MarchSetup& m = c->msetup;
  const size_t P = (size_t)m.npan, nfoil = P;
  const size_t row_doubles = kMarchRowHead + 2 * (size_t)m.ncoef + 2 * P;
  const long long n0 = (long long)c->wake_n;
  if ((long long)state[0] != n0) return fail(c, LUDVM_E_ARG, "march: state[0] must be the current wake size");
  if (n0 + 2 * count + (long long)nfoil >= (1LL << 32)) return fail(c, LUDVM_E_ARG, "march: wake too large");
  if (hist && (long long)hist_nmax < n0 + 2 * count) return fail(c, LUDVM_E_ARG, "march: history rows shorter than the wake can get");
  HIPCHK(c, hipSetDevice(c->device));
  // everything that could reallocate happens before the first launch: the steps then run without a host sync
  CHK(wake_grow(c, (size_t)(n0 + 2 * count) + nfoil));
  size_t part_bytes = 0, acc_bytes = 0;
  for (long long k = 0; k <= count; k += std::max<long long>(1, count / 256))
    march_workspace(c, n0 + 2 * k, precision, nfoil, part_bytes, acc_bytes);
  march_workspace(c, n0 + 2 * count, precision, nfoil, part_bytes, acc_bytes);
  CHK(ensure(c, c->part, part_bytes + (1 << 20)));
  if (acc_bytes && !c->ext_acc) CHK(ensure(c, c->acc, acc_bytes + (1 << 20)));
  if (acc_bytes && c->ext_acc && acc_bytes + 16 > c->ext_acc_bytes)
    return fail(c, LUDVM_E_NOMEM, "march: the accumulator buffer given to ludvm_set_shard is too small for this stretch");
  CHK(ensure(c, c->symsc, 128));
  CHK(ensure(c, c->march_rows, (size_t)count * row_doubles * 8));
  if (hist) CHK(ensure(c, c->march_hist, (size_t)count * 2 * hist_nmax * 8));

  MarchState hs{};
  hs.n = n0;
  hs.itev = (long long)state[1];
  hs.ilev = (long long)state[2];
  hs.shed = state[3] != 0.0;
  hs.tail = 0;
  hs.lesp_crit = state[4]; hs.sum_tev = state[5]; hs.sum_lev = state[6];
  for (int k = 0; k < 4; ++k) hs.place[k] = state[7 + k];
  for (int k = 0; k < m.ncoef; ++k) hs.prevA[k] = state[16 + k];
  MarchState* S = static_cast<MarchState*>(c->march_state.p);
  HIPCHK(c, hipMemcpyAsync(S, &hs, sizeof(MarchState), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // hs lives on this stack frame
  for (int k = 0; k < kProgressRing; ++k) c->progress[k] = 0;

  double* drows = static_cast<double*>(c->march_rows.p);
  const double* kin = static_cast<const double*>(c->march_kin.p);
  const size_t krow = 7 + 2 * P;
  hipLaunchKernelGGL(march_begin, dim3(1), dim3(kBlock), 0, c->stream, S, kin + (size_t)first_step * krow, (int)P,
                     (int)(first_step & 1), c->g64, m.vc4);
  HIPCHK(c, hipGetLastError());

  // The wake size is decided on the device (LEV shedding); the host needs an upper bound of it to size each step's
  // launches, and -- for the direct kernels -- that bound also fixes how the sources are split, i.e. the summation
  // order.  Every kSyncEvery steps the host waits for the event it recorded 2 * kSyncEvery steps earlier and reads,
  // from the progress ring, the wake size after the last step enqueued BEFORE that event: a step that is certainly
  // finished, and always the same one.  All bounds are therefore functions of the call's arguments and of the
  // simulation itself, never of how far the host happens to run ahead: two runs repeat bit for bit, direct or
  // symmetric kernel (whose fixed-point sums do not depend on the order of their atomics).
  //
  // Round 3: the bounds no longer restart at the call.  Step s is sized from the ANCHOR step A(s) = 64 (floor(s / 64) - 2) - 1
  // (the last step of the sync period two periods back; step 0 for the first 128 steps of a run): an anchor inside this
  // call is read from the ring as above, an anchor before it is given by the caller in `anchors` (the class knows the wake
  // size after every step it has run).  Tile size, waves per item, serial or overlapped step and the direct kernels'
  // source splits are then functions of the step number and of the simulation alone: the same bits whatever the chunking
  // (march_chunk, checkpoint_every, dense or sparse history) -- and a resumed run continues bit for bit.
  // Round 4 (ABI 4): FOUR anchors, periods floor(first_step / 64) - 3 + q, in an argument of their own.  The bound after
  // step first_step - 1 -- the first step's `n_before` -- belongs to the period before first_step's when first_step is a
  // multiple of 64, one period further back than the three that state[12..14] used to carry: every call that began at
  // 192, 256, 320, ... failed with "anchor step not among the caller's" (default chunks never start there,
  // snapshot_steps, checkpoint_every and resume do).  -1 = not given (0 is a wake size: a run without free vortices has
  // an empty wake after step 0); anchors = NULL or all four -1: the bounds restart at this call's exact wake size.
  constexpr long long kSyncEvery = 64;
  auto anchor_of = [](long long s) { return std::max<long long>(kSyncEvery * (s / kSyncEvery - 2) - 1, 0); };
  long long ev_anchor[2] = {-1, -1};      // the last step enqueued before the slot's event was recorded
  const long long k0 = first_step / kSyncEvery - 3;
  long long given_step[4], given_n[4];
  bool anchors_given = false;
  for (int q = 0; q < 4; ++q) {
    given_step[q] = std::max<long long>(kSyncEvery * (k0 + q) - 1, 0);
    given_n[q] = anchors ? anchors[q] : -1;
    if (given_n[q] < 0) continue;
    anchors_given = true;
    // a given size must be one the wake can have had: one or two vortices per step between the anchor and now
    const long long gap = first_step - 1 - given_step[q];
    if (gap < 0 || given_n[q] + gap > n0 || given_n[q] + 2 * gap < n0)
      return fail(c, LUDVM_E_ARG, "march: anchors[] (wake sizes after the anchor steps) contradict state[0]");
  }
  long long cur_a = -1, cur_n = 0;        // the anchor in force and the wake size after it
  // wake size after step s's solve is at most n_after(anchor) + 2 (s - anchor)
  auto set_anchor = [&](long long sstep) -> int {
    const long long a = anchor_of(sstep);
    if (!anchors_given && a < first_step) {
      // (no history from the caller: anchors before the call are replaced by the call's own start)
      cur_a = first_step - 1; cur_n = n0;
      return LUDVM_OK;
    }
    if (a == cur_a) return LUDVM_OK;
    if (a < first_step) {
      for (int q = 0; q < 4; ++q)
        if (given_step[q] == a && given_n[q] >= 0) { cur_a = a; cur_n = given_n[q]; return LUDVM_OK; }
      return fail(c, LUDVM_E_ARG, "march: the wake size after anchor step " + std::to_string(a) + " is not among anchors[]");
    }
    const int slot = (int)(((a + 1) / kSyncEvery) & 1);
    if (ev_anchor[slot] != a) return fail(c, LUDVM_E_STATE, "march: no event for the anchor step");
    HIPCHK(c, hipEventSynchronize(c->march_ev[slot]));
    const unsigned long long w = __atomic_load_n(c->progress + (a % kProgressRing), __ATOMIC_RELAXED);
    if ((long long)(w >> 32) != a) return fail(c, LUDVM_E_STATE, "march: progress ring out of step");
    cur_a = a;
    cur_n = (long long)(w & 0xffffffffULL);
    return LUDVM_OK;
  };
  CHK(set_anchor(first_step - 1 > 0 ? first_step - 1 : 0));
  long long prev_ub = cur_n + 2 * (std::max<long long>(first_step - 1, 0) - cur_a);   // bound after step first_step - 1
  if (prev_ub < n0) return fail(c, LUDVM_E_ARG, "march: anchors[] contradict state[0]");
  bool overlapped = false;          // the accumulators have been zeroed for the overlapped steps
  bool fork_signalled = false;      // the previous step's finisher already signals ev_fork
  static const bool ext_fork = [] { const char* e = LUDVM_EXP_ENV("LUDVM_MARCH_EXT_EVENTS"); return !(e && e[0] == '0'); }();
  // LUDVM_MARCH_OVERLAP=0 keeps every step serial (A/B measurements; results agree to fp32 rounding)
  const char* ov_env = LUDVM_EXP_ENV("LUDVM_MARCH_OVERLAP");
  const bool overlap_ok = !(ov_env && ov_env[0] == '0');
  hipStream_t const main_stream = c->stream;
  const double vc4 = m.vc4;
  const long long thr = sym_threshold(c, overlap_ok);      // serial symmetric steps pay from the usual size only
  for (long long s = first_step; s < first_step + count; ++s) {
    const long long rel = s - first_step;
    CHK(set_anchor(s));       // (may wait for, and read, the event slot that is re-used just below)
    if (s % kSyncEvery == 0 && s - 1 >= first_step) {
      // everything up to step s - 1 is enqueued: this event's completion makes s - 1 an anchor that can be read
      const int slot = (int)((s / kSyncEvery) & 1);
      HIPCHK(c, hipEventRecord(c->march_ev[slot], c->stream));
      ev_anchor[slot] = s - 1;
    }
    // wake size after this step's solve: at most two vortices per step since the anchor
    const long long n_ub = cur_n + 2 * (s - cur_a);
    const long long n_lo = cur_n + (s - 1 - cur_a);      // ... and before it: at least one per step
    long long n_before = std::min<long long>(prev_ub, n_ub);     // upper bound of the wake size before this step's solve
    prev_ub = n_ub;
    const bool symreg = precision != LUDVM_PREC_F64 && use_symmetric(c, n_ub, vc4, overlap_ok);
    // overlapped steps need an old wake that already fills the symmetric kernel
    const bool fork = symreg && overlap_ok && n_lo >= thr;
    const double* krow_s = kin + (size_t)s * krow;
    const double* krow_next = (size_t)(s + 1) < c->march_kin_rows ? kin + (size_t)(s + 1) * krow : nullptr;
    TailDuty td = make_tail_duty(S, s, krow_next, (int)P);
    if (hist) {
      td.hist_row = static_cast<double*>(c->march_hist.p) + (size_t)rel * 2 * hist_nmax;
      td.hist_nmax = (long long)hist_nmax;
    }
    double* row = drows + (size_t)rel * row_doubles;
    if (!fork) {
      // serial step: chord sums -> solve -> roll-up (direct, or symmetric with its memset) and Euler finisher
      CHK(march_chord_launch(c, n_before));
      hipLaunchKernelGGL(march_solve, dim3(1), dim3(kBlock), 0, c->stream, m, S, krow_s, row, s, c->x64, c->z64, c->g64,
                         c->mir(), c->g32, c->progress_dev);
      HIPCHK(c, hipGetLastError());
      MarchSym ms;
      ms.scale = &S->sc[(s + 1) & 1]; ms.bad = &S->sym_bad; ms.n_lo = n_lo + 1; ms.march = overlap_ok;
      CHK(advect_launch(c, (size_t)n_ub, &S->n, m.dt, nfoil, c->march_vcore, precision, nullptr, nullptr, td, ms));
      overlapped = false;    // the serial symmetric step leaves its sums in the accumulators
    } else {
      // overlapped step: the symmetric kernel on the wake as the last roll-up left it runs on the main stream
      // while chord sums and solve run on the second one; they meet at the Euler finisher
      const bool hilo = precision == LUDVM_PREC_F32X2;
      const long long nt_pad = (n_ub + 63) / 64 * 64;
      long long* acc = nullptr;
      CHK(acc_buffer(c, nt_pad, &acc));
      const bool sharded = sharded_at(c, n_ub);
      if (!overlapped) {
        // (march_finish_sym re-zeroes what it reads, so once is enough)
        const size_t all = c->ext_acc ? c->ext_acc_bytes : c->acc.cap;
        HIPCHK(c, hipMemsetAsync(acc - 2, 0, all, c->stream));
        overlapped = true;
      }
      // (an overlapped step's finisher carries the fork event as its completion signal: no packet of its own)
      if (!fork_signalled) HIPCHK(c, hipEventRecord(c->ev_fork, main_stream));
      HIPCHK(c, hipStreamWaitEvent(c->stream_b, c->ev_fork, 0));
      c->stream = c->stream_b;
      int rc = march_chord_launch(c, n_before);
      if (rc == LUDVM_OK) {
        hipLaunchKernelGGL(march_solve, dim3(1), dim3(kBlock), 0, c->stream, m, S, krow_s, row, s, c->x64, c->z64, c->g64,
                           c->mir(), c->g32, c->progress_dev);
        if (hipGetLastError() != hipSuccess) rc = fail(c, LUDVM_E_HIP, "march_solve launch failed");
      }
      c->stream = main_stream;
      CHK(rc);
      HIPCHK(c, hipEventRecord(c->ev_join, c->stream_b));
      const long long nb = std::max<long long>(n_before, 1);
      const int T = sym_tile_t(c, nb, hilo, !hilo);
      const long long ntiles = (nb + 64LL * T - 1) / (64LL * T);
      SymOperands o{};
      o.g = c->g32;
      if (hilo) { o.x = c->xh; o.z = c->zh; o.xl = c->xl; o.zl = c->zl; }
      else { o.x = c->xr; o.z = c->zr; o.cx = c->cx; o.cz = c->cz; }
      o.acc_u = acc; o.acc_w = acc + nt_pad;
      o.scale = &S->sc[s & 1];
      // sharded: this step's NaN counter travels with the sums through the all-reduce (two slots by step parity: the
      // finisher clears the next step's while blocks of its own launch may still read this step's)
      long long* bad_step = sharded ? acc - 2 + (s & 1) : nullptr;
      long long* bad_next = sharded ? acc - 2 + ((s + 1) & 1) : nullptr;
      o.bad = sharded ? bad_step : &S->sym_bad;
      long long first = 0, cnt = ntiles;
      if (sharded) shard_tiles(c, ntiles, &first, &cnt);
      CHK(launch_sym_tiles(c, T, o, nb, first, cnt, vc4, &S->n_old[s & 1], n_lo, sharded));
      if (sharded) CHK(reduce_accumulators(c, acc, nt_pad));
      HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
      if (ext_fork) {
        hipExtLaunchKernelGGL(march_finish_sym, dim3(fin_blocks(n_ub)), dim3(kFinBlock), 0, c->stream, nullptr, c->ev_fork, 0,
                              acc, acc + nt_pad, (const SymScale*)&S->sc[s & 1], S,
                              (const long long*)&S->n_old[s & 1], (int)nfoil, (float)vc4, m.dt, c->x64, c->z64, c->mir(), c->g32, td,
                              bad_step, bad_next);
        fork_signalled = true;
      } else {
        hipLaunchKernelGGL(march_finish_sym, dim3(fin_blocks(n_ub)), dim3(kFinBlock), 0, c->stream, acc, acc + nt_pad,
                           &S->sc[s & 1], S, &S->n_old[s & 1], (int)nfoil, (float)vc4, m.dt, c->x64, c->z64, c->mir(), c->g32, td,
                           bad_step, bad_next);
      }
      HIPCHK(c, hipGetLastError());
    }
    if (!fork) fork_signalled = false;
  }
  // results: per-step rows, final state, the two newest wake vortices
  HIPCHK(c, hipMemcpyAsync(rows, drows, (size_t)count * row_doubles * 8, hipMemcpyDeviceToHost, c->stream));
  if (hist) HIPCHK(c, hipMemcpyAsync(hist, c->march_hist.p, (size_t)count * 2 * hist_nmax * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&hs, S, sizeof(MarchState), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (hs.n < n0 + count || hs.n > n0 + 2 * count) return fail(c, LUDVM_E_STATE, "march: inconsistent wake size on the device");
  c->wake_n = (size_t)hs.n;
  state[0] = (double)hs.n; state[1] = (double)hs.itev; state[2] = (double)hs.ilev; state[3] = (double)hs.shed;
  state[4] = hs.lesp_crit; state[5] = hs.sum_tev; state[6] = hs.sum_lev;
  for (int k = 0; k < 4; ++k) state[7 + k] = hs.place[k];
  state[11] = (double)hs.tail;
  double tailbuf[4] = {0, 0, 0, 0};
  const size_t nn = (size_t)hs.n;
  if (nn >= 2) {
    HIPCHK(c, hipMemcpy(tailbuf, c->x64 + nn - 2, 16, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(tailbuf + 2, c->z64 + nn - 2, 16, hipMemcpyDeviceToHost));
  }
  for (int k = 0; k < 4; ++k) state[12 + k] = tailbuf[k];
  for (int k = 0; k < m.ncoef; ++k) state[16 + k] = hs.prevA[k];
  if (c->timing) CHK(drain_timing(c));
  return LUDVM_OK;
}

}  // extern "C"
