// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: the resident wake (float64 masters + fp32 mirrors on local origins) and its roll-up (LUDVM.py:1095-1127).
#include "ctx.hpp"
#include "wake_kernels.hpp"

namespace ludvm_host {

int wake_grow(ludvm_ctx* c, size_t capacity) {
  if (capacity <= c->wake_cap) return LUDVM_OK;
  size_t cap = std::max(capacity, std::max<size_t>(4096, c->wake_cap * 2));
  cap = (cap + kOriginBlock - 1) / kOriginBlock * kOriginBlock;      // whole origin blocks
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->stream_b) HIPCHK(c, hipStreamSynchronize(c->stream_b));
  const size_t nblk = (size_t)origin_slots((long long)cap);      // origin records: two per 256-vortex block
  double* d64[3] = {nullptr, nullptr, nullptr};
  float* f32[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipError_t me = hipSuccess;
  for (auto& q : d64) if (me == hipSuccess) me = hipMalloc(&q, cap * sizeof(double));
  for (int k = 0; k < 7; ++k) if (me == hipSuccess) me = hipMalloc(&f32[k], cap * sizeof(float));
  for (int k = 7; k < 9; ++k) if (me == hipSuccess) me = hipMalloc(&f32[k], nblk * sizeof(float));
  // (records of blocks that hold no vortex yet are never used for a stored vortex, but tiles that straddle the end of
  // the wake read them: they must be numbers)
  for (int k = 7; k < 9; ++k) if (me == hipSuccess) me = hipMemsetAsync(f32[k], 0, nblk * sizeof(float), c->stream);
  if (me != hipSuccess) {      // the wake keeps its old arrays; what was obtained so far goes back
    for (double* q : d64) if (q) (void)hipFree(q);
    for (float* q : f32) if (q) (void)hipFree(q);
    return fail_hip(c, "hipMalloc (wake arrays)", me);
  }
  const size_t n = c->wake_n;
  double* o64[3] = {c->x64, c->z64, c->g64};
  float* o32[9] = {c->xh, c->xl, c->zh, c->zl, c->g32, c->xr, c->zr, c->cx, c->cz};
  if (n) {
    for (int k = 0; k < 3; ++k) HIPCHK(c, hipMemcpyAsync(d64[k], o64[k], n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    for (int k = 0; k < 7; ++k) HIPCHK(c, hipMemcpyAsync(f32[k], o32[k], n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    const size_t ob = 2 * ((n + kOriginBlock - 1) / kOriginBlock);
    for (int k = 7; k < 9; ++k) HIPCHK(c, hipMemcpyAsync(f32[k], o32[k], ob * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  for (double* q : o64)
    if (q) HIPCHK(c, hipFree(q));
  for (float* q : o32)
    if (q) HIPCHK(c, hipFree(q));
  c->x64 = d64[0]; c->z64 = d64[1]; c->g64 = d64[2];
  c->xh = f32[0]; c->xl = f32[1]; c->zh = f32[2]; c->zl = f32[3]; c->g32 = f32[4];
  c->xr = f32[5]; c->zr = f32[6]; c->cx = f32[7]; c->cz = f32[8];
  c->wake_cap = cap;
  return LUDVM_OK;
}

// Rebuild the fp32 mirrors of [first, first + count) -- and of the rest of the origin blocks they touch -- from the
// float64 masters.
int wake_refresh(ludvm_ctx* c, size_t first, size_t count) {
  if (!count) return LUDVM_OK;
  const long long lo = (long long)(first / kOriginBlock * kOriginBlock);
  const long long hi = std::min<long long>((long long)c->wake_cap, (long long)((first + count + kOriginBlock - 1) / kOriginBlock * kOriginBlock));
  hipLaunchKernelGGL(refresh_mirrors, dim3(blocks_for(hi - lo)), dim3(kBlock), 0, c->stream, (long long)first, (long long)count,
                     (long long)std::max(c->wake_n, first + count), (long long)c->wake_cap, c->x64, c->z64, c->g64, c->mir(), c->g32);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

int advect_launch(ludvm_ctx* c, size_t n, const long long* n_dev, double dt, size_t nfoil, double vcore, int precision,
                  double* du, double* dw, TailDuty td, MarchSym ms) {
  const long long ns = (long long)(n + nfoil), nt = (long long)n;
  const double v2 = vcore * vcore;
  const bool hilo = precision == LUDVM_PREC_F32X2;
  if (precision != LUDVM_PREC_F64 && use_symmetric(c, nt, v2 * v2, ms.march)) {
    // wake x wake: each unordered pair once; the bound vortices' part is summed in the Euler finisher
    long long nt_pad = 0;
    SymOperands o{};
    o.g = c->g32;
    if (hilo) { o.x = c->xh; o.z = c->zh; o.xl = c->xl; o.zl = c->zl; }
    else { o.x = c->xr; o.z = c->zr; o.cx = c->cx; o.cz = c->cz; }
    o.scale = ms.scale; o.bad = ms.bad;
    const long long *acc = nullptr, *bad = nullptr;
    CHK(launch_sym(c, o, nt, v2 * v2, &nt_pad, &acc, &bad, n_dev, ms.n_lo));
    hipLaunchKernelGGL(finish_wake_advect_sym, dim3(fin_blocks(nt)), dim3(kFinBlock), 0, c->stream, acc, acc + nt_pad,
                       ms.scale ? ms.scale : ctx_scale(c), bad, nt, (int)nfoil, (float)(v2 * v2), dt,
                       c->x64, c->z64, c->mir(), c->g32, du, dw, n_dev, td);
    HIPCHK(c, hipGetLastError());
    return LUDVM_OK;
  }
  PairArgs a{};
  a.ns = ns;
  a.nt = nt;
  if (n_dev) {          // sizes relative to the device-side wake size
    a.n_dev = n_dev; a.ns_dev = 1; a.nt_dev = 1;
    a.ns = (long long)nfoil;
    a.nt = 0;
  }
  a.vc4 = v2 * v2;
  if (precision == LUDVM_PREC_F64) {
    a.xs = c->x64; a.zs = c->z64; a.gs = c->g64; a.xt = c->x64; a.zt = c->z64;
  } else if (hilo) {
    a.xs = c->xh; a.zs = c->zh; a.gs = c->g32; a.xsl = c->xl; a.zsl = c->zl;
    a.xt = c->xh; a.zt = c->zh; a.xtl = c->xl; a.ztl = c->zl;
  } else {
    // fp32 with local origins: offsets from the origin of each 256-vortex block of the wake array
    a.xs = c->xr; a.zs = c->zr; a.gs = c->g32; a.scx = c->cx; a.scz = c->cz;
    a.xt = c->xr; a.zt = c->zr; a.tcx = c->cx; a.tcz = c->cz; a.t_index0 = 0;
  }
  Plan p = make_plan(c, nt, ns, precision);
  CHK(launch_pair(c, a, p, precision, nullptr, nullptr));  // results stay in the slab
  if (precision == LUDVM_PREC_F64)
    hipLaunchKernelGGL(finish_wake_advect<double>, dim3(fin_blocks(nt)), dim3(kFinBlock), 0, c->stream,
                       static_cast<const double*>(c->part.p), nt, p.nt_pad, p.nsplit, dt, c->x64, c->z64, c->mir(), du, dw,
                       n_dev, td);
  else
    hipLaunchKernelGGL(finish_wake_advect<float>, dim3(fin_blocks(nt)), dim3(kFinBlock), 0, c->stream,
                       static_cast<const float*>(c->part.p), nt, p.nt_pad, p.nsplit, dt, c->x64, c->z64, c->mir(), du, dw,
                       n_dev, td);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

}  // namespace ludvm_host

extern "C" {

/* ---- resident wake ------------------------------------------------------------------------ */

int ludvm_wake_reserve(ludvm_ctx* c, size_t capacity) {
  if (!c) return LUDVM_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  return wake_grow(c, capacity);
}

int ludvm_wake_clear(ludvm_ctx* c) {
  if (!c) return LUDVM_E_ARG;
  c->wake_n = 0;
  return LUDVM_OK;
}

int ludvm_wake_size(ludvm_ctx* c, size_t* n) {
  if (!c || !n) return LUDVM_E_ARG;
  *n = c->wake_n;
  return LUDVM_OK;
}

int ludvm_wake_truncate(ludvm_ctx* c, size_t n) {
  if (!c) return LUDVM_E_ARG;
  if (n > c->wake_n) return fail(c, LUDVM_E_ARG, "truncate beyond the wake size");
  c->wake_n = n;
  return LUDVM_OK;
}

int ludvm_wake_append(ludvm_ctx* c, const double* x, const double* z, const double* gamma, size_t count) {
  if (!c) return LUDVM_E_ARG;
  if (count == 0) return LUDVM_OK;
  if (!x || !z || !gamma) return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  CHK(wake_grow(c, c->wake_n + count));
  const size_t n = c->wake_n;
  CHK(h2d(c, c->x64 + n, x, count * 8));
  CHK(h2d(c, c->z64 + n, z, count * 8));
  CHK(h2d(c, c->g64 + n, gamma, count * 8));
  CHK(wake_refresh(c, n, count));
  c->wake_n = n + count;
  return LUDVM_OK;
}

int ludvm_wake_write(ludvm_ctx* c, size_t first, size_t count, const double* x, const double* z, const double* gamma) {
  if (!c) return LUDVM_E_ARG;
  if (first + count > c->wake_n) return fail(c, LUDVM_E_ARG, "range outside the wake");
  if (count == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (x) CHK(h2d(c, c->x64 + first, x, count * 8));
  if (z) CHK(h2d(c, c->z64 + first, z, count * 8));
  if (gamma) CHK(h2d(c, c->g64 + first, gamma, count * 8));
  return wake_refresh(c, first, count);
}

int ludvm_wake_read(ludvm_ctx* c, size_t first, size_t count, double* x, double* z, double* gamma) {
  if (!c) return LUDVM_E_ARG;
  if (first + count > c->wake_n) return fail(c, LUDVM_E_ARG, "range outside the wake");
  if (count == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (x) HIPCHK(c, hipMemcpyAsync(x, c->x64 + first, count * 8, hipMemcpyDeviceToHost, c->stream));
  if (z) HIPCHK(c, hipMemcpyAsync(z, c->z64 + first, count * 8, hipMemcpyDeviceToHost, c->stream));
  if (gamma) HIPCHK(c, hipMemcpyAsync(gamma, c->g64 + first, count * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

int ludvm_wake_induce_on_points(ludvm_ctx* c, size_t src_first, size_t src_count, const double* xt, const double* zt,
                                size_t nt, double vcore, double* u, double* w) {
  if (!c) return LUDVM_E_ARG;
  if (src_first + src_count > c->wake_n) return fail(c, LUDVM_E_ARG, "source range outside the wake");
  if (nt && (!xt || !zt || !u || !w)) return fail(c, LUDVM_E_ARG, "null array");
  if (nt == 0) return LUDVM_OK;
  if (src_count == 0) {
    std::memset(u, 0, nt * sizeof(double));
    std::memset(w, 0, nt * sizeof(double));
    return LUDVM_OK;
  }
  HIPCHK(c, hipSetDevice(c->device));
  CHK(ensure(c, c->arena, 4 * Arena::need(nt, 8)));
  Arena ar(c->arena.p);
  double* dxt = ar.take<double>(nt);
  double* dzt = ar.take<double>(nt);
  double* du = ar.take<double>(nt);
  double* dw = ar.take<double>(nt);
  HIPCHK(c, hipMemcpyAsync(dxt, xt, nt * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dzt, zt, nt * 8, hipMemcpyHostToDevice, c->stream));
  PairArgs a{};
  a.xs = c->x64 + src_first; a.zs = c->z64 + src_first; a.gs = c->g64 + src_first;
  a.ns = (long long)src_count;
  a.xt = dxt; a.zt = dzt; a.nt = (long long)nt;
  const double v2 = vcore * vcore;
  a.vc4 = v2 * v2;
  CHK(induce_device(c, a, (long long)nt, (long long)src_count, LUDVM_PREC_F64, du, dw));
  HIPCHK(c, hipMemcpyAsync(u, du, nt * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(w, dw, nt * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

int ludvm_wake_chord_sums(ludvm_ctx* c, size_t src_first, size_t src_count, const double* xt, const double* zt, size_t nt,
                          const double* unit_x, const double* unit_z, size_t n_unit, double vcore, double* u_wake,
                          double* w_wake, double* u_unit, double* w_unit) {
  if (!c) return LUDVM_E_ARG;
  if (src_first + src_count > c->wake_n) return fail(c, LUDVM_E_ARG, "source range outside the wake");
  if (nt == 0) return LUDVM_OK;
  if (!xt || !zt || !u_wake || !w_wake) return fail(c, LUDVM_E_ARG, "null array");
  if (n_unit > 4) return fail(c, LUDVM_E_ARG, "at most 4 unit vortices");
  if (n_unit && (!unit_x || !unit_z || !u_unit || !w_unit)) return fail(c, LUDVM_E_ARG, "null unit array");
  const size_t in_doubles = 2 * nt + 2 * n_unit, out_doubles = 2 * nt * (1 + n_unit);
  if (out_doubles * 8 > kPinOutBytes || in_doubles * 8 > kPinBytes / 4) {
    // large point sets: the general entry points
    CHK(ludvm_wake_induce_on_points(c, src_first, src_count, xt, zt, nt, vcore, u_wake, w_wake));
    const double one = 1.0;
    for (size_t k = 0; k < n_unit; ++k)
      CHK(ludvm_induce_f64(c, unit_x + k, unit_z + k, &one, 1, xt, zt, nt, vcore, LUDVM_PREC_F64, u_unit + k * nt,
                           w_unit + k * nt));
    return LUDVM_OK;
  }
  HIPCHK(c, hipSetDevice(c->device));
  CHK(ensure(c, c->arena, Arena::need(in_doubles, 8) + Arena::need(out_doubles, 8)));
  Arena ar(c->arena.p);
  double* din = ar.take<double>(in_doubles);
  double* dout = ar.take<double>(out_doubles);
  std::vector<double> pack(in_doubles);
  std::memcpy(pack.data(), xt, nt * 8);
  std::memcpy(pack.data() + nt, zt, nt * 8);
  if (n_unit) {
    std::memcpy(pack.data() + 2 * nt, unit_x, n_unit * 8);
    std::memcpy(pack.data() + 2 * nt + n_unit, unit_z, n_unit * 8);
  }
  CHK(h2d(c, din, pack.data(), in_doubles * 8));
  const double v2 = vcore * vcore;
  if (src_count) {
    PairArgs a{};
    a.xs = c->x64 + src_first; a.zs = c->z64 + src_first; a.gs = c->g64 + src_first;
    a.ns = (long long)src_count;
    a.xt = din; a.zt = din + nt; a.nt = (long long)nt;
    a.vc4 = v2 * v2;
    CHK(induce_device(c, a, (long long)nt, (long long)src_count, LUDVM_PREC_F64, dout, dout + nt));
  } else {
    HIPCHK(c, hipMemsetAsync(dout, 0, 2 * nt * 8, c->stream));
  }
  if (n_unit) {
    hipLaunchKernelGGL(unit_influence_f64, dim3(blocks_for((long long)(nt * n_unit))), dim3(kBlock), 0, c->stream, din,
                       din + nt, (long long)nt, din + 2 * nt, din + 2 * nt + n_unit, (int)n_unit, v2 * v2, dout + 2 * nt);
    HIPCHK(c, hipGetLastError());
  }
  void* hv = nullptr;
  CHK(d2h_small_sync(c, dout, out_doubles * 8, &hv));
  const double* h = static_cast<const double*>(hv);
  std::memcpy(u_wake, h, nt * 8);
  std::memcpy(w_wake, h + nt, nt * 8);
  for (size_t k = 0; k < n_unit; ++k) {
    std::memcpy(u_unit + k * nt, h + 2 * nt + (2 * k) * nt, nt * 8);
    std::memcpy(w_unit + k * nt, h + 2 * nt + (2 * k + 1) * nt, nt * 8);
  }
  return LUDVM_OK;
}

int ludvm_wake_advect_tail(ludvm_ctx* c, double dt, const double* foil_x, const double* foil_z, const double* foil_dgamma,
                           size_t nfoil, double vcore, int precision, size_t tail_count, double* tail_x, double* tail_z) {
  if (!c) return LUDVM_E_ARG;
  if (tail_count > c->wake_n) return fail(c, LUDVM_E_ARG, "tail longer than the wake");
  if (tail_count && (!tail_x || !tail_z)) return fail(c, LUDVM_E_ARG, "null tail array");
  CHK(ludvm_wake_advect(c, dt, foil_x, foil_z, foil_dgamma, nfoil, vcore, precision, nullptr, nullptr));
  if (tail_count == 0) return LUDVM_OK;
  if (2 * tail_count * 8 > kPinOutBytes) return ludvm_wake_read(c, c->wake_n - tail_count, tail_count, tail_x, tail_z, nullptr);
  // x and z tails are separate device ranges: stage them next to each other, then one read-back
  CHK(ensure(c, c->arena, Arena::need(2 * tail_count, 8)));
  double* stage = static_cast<double*>(c->arena.p);
  const size_t first = c->wake_n - tail_count;
  HIPCHK(c, hipMemcpyAsync(stage, c->x64 + first, tail_count * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(stage + tail_count, c->z64 + first, tail_count * 8, hipMemcpyDeviceToDevice, c->stream));
  void* hv = nullptr;
  CHK(d2h_small_sync(c, stage, 2 * tail_count * 8, &hv));
  std::memcpy(tail_x, hv, tail_count * 8);
  std::memcpy(tail_z, static_cast<const double*>(hv) + tail_count, tail_count * 8);
  return LUDVM_OK;
}

int ludvm_wake_advect(ludvm_ctx* c, double dt, const double* foil_x, const double* foil_z, const double* foil_dgamma,
                      size_t nfoil, double vcore, int precision, double* u_out, double* w_out) {
  if (!c) return LUDVM_E_ARG;
  if (!valid_precision(precision)) return fail(c, LUDVM_E_ARG, "unknown precision");
  if (nfoil && (!foil_x || !foil_z || !foil_dgamma)) return fail(c, LUDVM_E_ARG, "null foil array");
  if ((u_out == nullptr) != (w_out == nullptr)) return fail(c, LUDVM_E_ARG, "u_out and w_out go together");
  const size_t n = c->wake_n;
  if (n == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  // bound vortices ride behind the wake in the same source arrays for this launch
  CHK(wake_grow(c, n + nfoil));
  if (nfoil) {
    CHK(h2d(c, c->x64 + n, foil_x, nfoil * 8));
    CHK(h2d(c, c->z64 + n, foil_z, nfoil * 8));
    CHK(h2d(c, c->g64 + n, foil_dgamma, nfoil * 8));
    CHK(wake_refresh(c, n, nfoil));
  }
  double *du = nullptr, *dw = nullptr;
  if (u_out) {
    CHK(ensure(c, c->arena, 2 * Arena::need(n, 8)));
    Arena ar(c->arena.p);
    du = ar.take<double>(n);
    dw = ar.take<double>(n);
  }
  CHK(advect_launch(c, n, nullptr, dt, nfoil, vcore, precision, du, dw));
  if (u_out) {
    HIPCHK(c, hipMemcpyAsync(u_out, du, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(w_out, dw, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LUDVM_OK;
}

int ludvm_wake_step(ludvm_ctx* c, const double* new_x, const double* new_z, const double* new_gamma, size_t n_new, double dt,
                    const double* foil_x, const double* foil_z, const double* foil_dgamma, size_t nfoil, double vcore,
                    int precision, const double* te, const double* le, int lev_from_prev, size_t tail_count,
                    const double* xt, const double* zt, size_t nt, double* tail_x, double* tail_z, double* unit_x,
                    double* unit_z, double* u_wake, double* w_wake, double* u_unit, double* w_unit) {
  if (!c) return LUDVM_E_ARG;
  if (!valid_precision(precision)) return fail(c, LUDVM_E_ARG, "unknown precision");
  if (n_new && (!new_x || !new_z || !new_gamma)) return fail(c, LUDVM_E_ARG, "null new-vortex array");
  if (nfoil && (!foil_x || !foil_z || !foil_dgamma)) return fail(c, LUDVM_E_ARG, "null foil array");
  const size_t n0 = c->wake_n, n = n0 + n_new;
  if (tail_count < 1 || tail_count > 2 || tail_count > n) return fail(c, LUDVM_E_ARG, "tail_count must be 1 or 2");
  if (!te || !le || !xt || !zt || !tail_x || !tail_z || !unit_x || !unit_z || !u_wake || !w_wake || !u_unit || !w_unit || nt == 0)
    return fail(c, LUDVM_E_ARG, "null array");
  const size_t n_stage = 3 * (n_new + nfoil);
  const size_t in_doubles = n_stage + 2 * nt + 4;           // staged vortices | xt | zt | te, le
  const size_t out_doubles = 2 * tail_count + 4 + 6 * nt;   // tail x|z, unit[4], wake u|w, unit u,w rows
  if (out_doubles * 8 > kPinOutBytes || in_doubles * 8 > kPinBytes / 4) return fail(c, LUDVM_E_ARG, "step too large for the fused call");
  HIPCHK(c, hipSetDevice(c->device));
  CHK(wake_grow(c, n + nfoil));
  CHK(ensure(c, c->arena, Arena::need(in_doubles, 8) + Arena::need(out_doubles, 8)));
  Arena ar(c->arena.p);
  double* din = ar.take<double>(in_doubles);
  double* dout = ar.take<double>(out_doubles);
  // ONE upload for the whole step
  std::vector<double>& pk = c->pack;
  pk.resize(in_doubles);
  double* q = pk.data();
  if (n_new) { std::memcpy(q, new_x, n_new * 8); std::memcpy(q + n_new, new_z, n_new * 8); std::memcpy(q + 2 * n_new, new_gamma, n_new * 8); }
  q += 3 * n_new;
  if (nfoil) { std::memcpy(q, foil_x, nfoil * 8); std::memcpy(q + nfoil, foil_z, nfoil * 8); std::memcpy(q + 2 * nfoil, foil_dgamma, nfoil * 8); }
  q += 3 * nfoil;
  std::memcpy(q, xt, nt * 8);
  std::memcpy(q + nt, zt, nt * 8);
  q[2 * nt] = te[0]; q[2 * nt + 1] = te[1]; q[2 * nt + 2] = le[0]; q[2 * nt + 3] = le[1];
  CHK(h2d(c, din, pk.data(), in_doubles * 8));
  if (n_new + nfoil) {
    hipLaunchKernelGGL(stage_step_inputs, dim3(blocks_for((long long)(n_new + nfoil))), dim3(kBlock), 0, c->stream, din,
                       (long long)n0, (int)n_new, (int)nfoil, c->x64, c->z64, c->g64, c->mir(), c->g32);
    HIPCHK(c, hipGetLastError());
  }
  c->wake_n = n;
  CHK(advect_launch(c, c->wake_n, nullptr, dt, nfoil, vcore, precision, nullptr, nullptr));
  const double* d_xt = din + n_stage;
  const double* d_zt = d_xt + nt;
  const double* d_geo = d_zt + nt;
  double* d_unit = dout + 2 * tail_count;      // [tev_x, lev_x, tev_z, lev_z]
  double* d_sums = d_unit + 4;                 // u_wake | w_wake | unit rows
  // fp64 wake -> chord partial sums stay in the slab; one small kernel then sums the splits, places the next
  // TEV / candidate LEV from the advected positions and evaluates their unit influences
  const double v2 = vcore * vcore;
  PairArgs a{};
  a.xs = c->x64; a.zs = c->z64; a.gs = c->g64; a.ns = (long long)n;
  a.xt = d_xt; a.zt = d_zt; a.nt = (long long)nt;
  a.vc4 = v2 * v2;
  Plan p = make_plan(c, (long long)nt, (long long)n, LUDVM_PREC_F64);
  const bool was = c->timing;
  c->timing = false;   // the chord sums are not the dominant kernel
  int rc = launch_pair(c, a, p, LUDVM_PREC_F64, nullptr, nullptr);   // results stay in c->part
  c->timing = was;
  CHK(rc);
  const double* slab = static_cast<const double*>(c->part.p);
  hipLaunchKernelGGL(chord_finish_f64, dim3(blocks_for((long long)(2 * nt * 64))), dim3(kBlock), 0, c->stream,
                     p.nsplit > 1 ? slab : (const double*)nullptr, p.nt_pad, p.nsplit, slab, d_xt, d_zt, (long long)nt, c->x64,
                     c->z64, (long long)n, (int)tail_count, lev_from_prev, d_geo, v2 * v2, dout, d_sums);
  HIPCHK(c, hipGetLastError());
  void* hv = nullptr;
  CHK(d2h_small_sync(c, dout, out_doubles * 8, &hv));
  const double* h = static_cast<const double*>(hv);
  std::memcpy(tail_x, h, tail_count * 8);
  std::memcpy(tail_z, h + tail_count, tail_count * 8);
  const double* hu = h + 2 * tail_count;
  unit_x[0] = hu[0]; unit_x[1] = hu[1]; unit_z[0] = hu[2]; unit_z[1] = hu[3];
  const double* hs = hu + 4;
  std::memcpy(u_wake, hs, nt * 8);
  std::memcpy(w_wake, hs + nt, nt * 8);
  for (size_t k = 0; k < 2; ++k) {
    std::memcpy(u_unit + k * nt, hs + 2 * nt + (2 * k) * nt, nt * 8);
    std::memcpy(w_unit + k * nt, hs + 2 * nt + (2 * k + 1) * nt, nt * 8);
  }
  return LUDVM_OK;
}

}  // extern "C"
