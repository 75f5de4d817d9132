// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: the stateless pair sums (LUDVM.induced_velocity, LUDVM.py:549-570) and the device-pointer entry points of the
// multi-GPU shard step.
#include "ctx.hpp"
#include "induce_kernels.hpp"

extern "C" {

/* ---- stateless pair sum ------------------------------------------------------------------- */

int ludvm_induce_f64(ludvm_ctx* c, const double* xs, const double* zs, const double* gs, size_t ns, const double* xt,
                     const double* zt, size_t nt, double vcore, int precision, double* u, double* w) {
  if (!c) return LUDVM_E_ARG;
  if (!valid_precision(precision)) return fail(c, LUDVM_E_ARG, "unknown precision");
  if ((ns && (!xs || !zs || !gs)) || (nt && (!xt || !zt || !u || !w))) return fail(c, LUDVM_E_ARG, "null array");
  if (nt == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (ns == 0) {
    std::memset(u, 0, nt * sizeof(double));
    std::memset(w, 0, nt * sizeof(double));
    return LUDVM_OK;
  }
  // fp32 on local origins needs compact origin classes on BOTH sides.  A side of fewer than kOrderMin points cannot be
  // made compact by ordering it (a class is 128 points whatever their number).  While the call is small as a whole
  // (<= 2^28 pairs: latency-bound, float64 costs ~0.2 ms at most) it runs in float64, whose accuracy does not depend on
  // the order (the G1 clouds of 257 x 1023 random points: 5e-4 ... 2e-3 of max|u| in fp32 before, rounding now).  A SMALL
  // side against a LARGE one (a few probe points in a wake of 1e6 ... 8e6 vortices) would pay the float64 rate on every
  // pair: it takes hi+lo positions instead (exact differences, no classes needed, 1.3 x the fp32 time, <= 2e-6 of max|u|)
  // (ADVICE r4).
  if (precision == LUDVM_PREC_F32 && std::min(ns, nt) < kOrderMin)
    precision = (double)ns * (double)nt <= kSmallSidePairsF64 ? LUDVM_PREC_F64 : LUDVM_PREC_F32X2;
  const bool f64 = precision == LUDVM_PREC_F64;
  bool hilo = precision == LUDVM_PREC_F32X2;
  // the caller passed the same arrays as sources and targets: self-interaction (the targets are not uploaded twice,
  // and from kSymMinN vortices the symmetric kernel takes it)
  const bool self = xt == xs && zt == zs && nt == ns;
  const size_t ntu = self ? 0 : nt;                       // targets uploaded
  const size_t in_doubles = 3 * ns + 2 * ntu, out_doubles = 2 * nt;
  // One packed block in = xs | zs | gs | xt | zt.  Small calls (every call of a README-size run) go through the pinned
  // ring: one upload, one conversion launch, the pair launch, one back-conversion, one pinned download.
  const bool small = in_doubles * 8 <= kPinBytes / 4 && out_doubles * 8 <= kPinOutBytes;
  const bool may_order = !f64 && !hilo;                   // (both sides >= kOrderMin then)
  const size_t nsb = (size_t)origin_slots((long long)ns), ntb = (size_t)origin_slots((long long)nt);
  size_t bytes = Arena::need(in_doubles, 8) + Arena::need(out_doubles, 8);
  if (!f64) bytes += 5 * Arena::need(ns, 4) + 6 * Arena::need(nt, 4) + 2 * Arena::need(nsb, 4) + 2 * Arena::need(ntb, 4);
  if (may_order) bytes += Arena::need(in_doubles, 8) + Arena::need(out_doubles, 8);      // the re-ordered copies
  CHK(ensure(c, c->arena, bytes));
  Arena ar(c->arena.p);
  double* din = ar.take<double>(in_doubles);
  double* dout = ar.take<double>(out_doubles);
  double* din_ord = may_order ? ar.take<double>(in_doubles) : nullptr;
  double* dout_ord = may_order ? ar.take<double>(out_doubles) : nullptr;
  if (small) {
    std::vector<double>& pk = c->pack;
    pk.resize(in_doubles);
    std::memcpy(pk.data(), xs, ns * 8);
    std::memcpy(pk.data() + ns, zs, ns * 8);
    std::memcpy(pk.data() + 2 * ns, gs, ns * 8);
    if (ntu) {
      std::memcpy(pk.data() + 3 * ns, xt, nt * 8);
      std::memcpy(pk.data() + 3 * ns + nt, zt, nt * 8);
    }
    CHK(h2d(c, din, pk.data(), in_doubles * 8));
  } else {
    HIPCHK(c, hipMemcpyAsync(din, xs, ns * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(din + ns, zs, ns * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(din + 2 * ns, gs, ns * 8, hipMemcpyHostToDevice, c->stream));
    if (ntu) {
      HIPCHK(c, hipMemcpyAsync(din + 3 * ns, xt, nt * 8, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(din + 3 * ns + nt, zt, nt * 8, hipMemcpyHostToDevice, c->stream));
    }
  }
  // unordered inputs (a caller's array, a turbulence cloud): sources, and targets that are not the sources, in Morton
  // order where that makes the origin classes compact; the results go back to the caller's order at the end
  const unsigned* ord_t = nullptr;
  if (may_order) {
    OrderWs ow{};
    CHK(order_workspace(c, std::max(ns, nt), &ow));
    const unsigned* ord_s = nullptr;
    double ext_s = 0.0, ext_t = 0.0;
    CHK(spatial_order_if_needed(c, ow, 0, din, din + ns, ns, &ord_s, &ext_s));
    if (self) ord_t = ord_s;
    else CHK(spatial_order_if_needed(c, ow, 1, din + 3 * ns, din + 3 * ns + nt, nt, &ord_t, &ext_t));
    if (too_sparse(ext_s, ord_s != nullptr, vcore) || too_sparse(ext_t, ord_t != nullptr && !self, vcore)) {
      hilo = true;                 // too sparse for its core: exact differences instead of an order (see too_sparse)
      ord_t = nullptr;
    } else if (ord_s || (ord_t && !self)) {
      CHK(order_gather(c, din, din + ns, din + 2 * ns, ord_s, (long long)ns, din_ord, din_ord + ns, din_ord + 2 * ns));
      if (!self)
        CHK(order_gather(c, din + 3 * ns, din + 3 * ns + nt, nullptr, ord_t, (long long)nt, din_ord + 3 * ns, din_ord + 3 * ns + nt,
                         nullptr));
      din = din_ord;
    }
  }
  PairArgs a{};
  a.ns = (long long)ns;
  a.nt = (long long)nt;
  const double v2 = vcore * vcore;
  a.vc4 = v2 * v2;
  if (f64) {
    a.xs = din; a.zs = din + ns; a.gs = din + 2 * ns;
    a.xt = self ? din : din + 3 * ns; a.zt = self ? din + ns : din + 3 * ns + nt;
    CHK(induce_device(c, a, (long long)nt, (long long)ns, precision, dout, dout + nt));
  } else {
    float* fxs = ar.take<float>(ns);
    float* fxsl = ar.take<float>(ns);     // lo parts (f32x2)
    float* fzs = ar.take<float>(ns);
    float* fzsl = ar.take<float>(ns);
    float* fgs = ar.take<float>(ns);
    float* fxt = ar.take<float>(nt);
    float* fxtl = ar.take<float>(nt);
    float* fzt = ar.take<float>(nt);
    float* fztl = ar.take<float>(nt);
    float* fu = ar.take<float>(nt);
    float* fw = ar.take<float>(nt);
    float* sox = ar.take<float>(nsb);     // block origins (f32: local-origin positions)
    float* soz = ar.take<float>(nsb);
    float* tox = ar.take<float>(ntb);
    float* toz = ar.take<float>(ntb);
    if (hilo)
      hipLaunchKernelGGL(cvt_packed_inputs, dim3(blocks_for((long long)in_doubles)), dim3(kBlock), 0, c->stream, din,
                         (long long)ns, (long long)ntu, fxs, fxsl, fzs, fzsl, fgs, fxt, fxtl, fzt, fztl);
    else
      hipLaunchKernelGGL(cvt_packed_inputs_local, dim3(blocks_for((long long)in_doubles)), dim3(kBlock), 0, c->stream, din,
                         (long long)ns, (long long)ntu, fxs, fzs, fgs, sox, soz, fxt, fzt, tox, toz);
    HIPCHK(c, hipGetLastError());
    if (self && use_symmetric(c, (long long)ns, a.vc4)) {
      long long nt_pad = 0;
      SymOperands o{};
      o.x = fxs; o.z = fzs; o.g = fgs;
      if (hilo) { o.xl = fxsl; o.zl = fzsl; } else { o.cx = sox; o.cz = soz; }
      const long long *acc = nullptr, *bad = nullptr;
      CHK(launch_sym(c, o, (long long)ns, a.vc4, &nt_pad, &acc, &bad));
      hipLaunchKernelGGL(finish_sym, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream, acc, acc + nt_pad, ctx_scale(c),
                         bad, (long long)nt, fu, fw);
      HIPCHK(c, hipGetLastError());
    } else {
      a.xs = fxs; a.zs = fzs; a.gs = fgs;
      a.xt = self ? fxs : fxt; a.zt = self ? fzs : fzt;
      if (hilo) {
        a.xsl = fxsl; a.zsl = fzsl;
        a.xtl = self ? fxsl : fxtl; a.ztl = self ? fzsl : fztl;
      } else {
        a.scx = sox; a.scz = soz;
        a.tcx = self ? sox : tox; a.tcz = self ? soz : toz;
        a.t_index0 = 0;
      }
      CHK(induce_device(c, a, (long long)nt, (long long)ns, hilo ? LUDVM_PREC_F32X2 : LUDVM_PREC_F32, fu, fw));
    }
    hipLaunchKernelGGL(cvt_packed_outputs, dim3(blocks_for((long long)out_doubles)), dim3(kBlock), 0, c->stream, fu, fw, dout,
                       (long long)nt);
    HIPCHK(c, hipGetLastError());
    if (ord_t) {
      CHK(order_scatter(c, dout, dout + nt, ord_t, (long long)nt, dout_ord, dout_ord + nt));
      dout = dout_ord;
    }
  }
  if (small) {
    void* hv = nullptr;
    CHK(d2h_small_sync(c, dout, out_doubles * 8, &hv));
    std::memcpy(u, hv, nt * 8);
    std::memcpy(w, static_cast<const double*>(hv) + nt, nt * 8);
    return LUDVM_OK;
  }
  HIPCHK(c, hipMemcpyAsync(u, dout, nt * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(w, dout + nt, nt * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

int ludvm_induce_f32(ludvm_ctx* c, const float* xs, const float* zs, const float* gs, size_t ns, const float* xt,
                     const float* zt, size_t nt, float vcore, float* u, float* w) {
  if (!c) return LUDVM_E_ARG;
  if ((ns && (!xs || !zs || !gs)) || (nt && (!xt || !zt || !u || !w))) return fail(c, LUDVM_E_ARG, "null array");
  if (nt == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (ns == 0) {
    std::memset(u, 0, nt * sizeof(float));
    std::memset(w, 0, nt * sizeof(float));
    return LUDVM_OK;
  }
  CHK(ensure(c, c->arena, 3 * Arena::need(ns, 4) + 4 * Arena::need(nt, 4)));
  Arena ar(c->arena.p);
  float* dxs = ar.take<float>(ns);
  float* dzs = ar.take<float>(ns);
  float* dgs = ar.take<float>(ns);
  float* dxt = ar.take<float>(nt);
  float* dzt = ar.take<float>(nt);
  float* du = ar.take<float>(nt);
  float* dw = ar.take<float>(nt);
  HIPCHK(c, hipMemcpyAsync(dxs, xs, ns * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dzs, zs, ns * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dgs, gs, ns * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dxt, xt, nt * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dzt, zt, nt * 4, hipMemcpyHostToDevice, c->stream));
  CHK(ludvm_induce_dev_f32(c, dxs, dzs, dgs, ns, dxt, dzt, nt, vcore, du, dw));
  HIPCHK(c, hipMemcpyAsync(u, du, nt * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(w, dw, nt * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

int ludvm_induce_dev_f32(ludvm_ctx* c, const float* d_xs, const float* d_zs, const float* d_gs, size_t ns,
                         const float* d_xt, const float* d_zt, size_t nt, float vcore, float* d_u, float* d_w) {
  if (!c) return LUDVM_E_ARG;
  if ((ns && (!d_xs || !d_zs || !d_gs)) || (nt && (!d_xt || !d_zt || !d_u || !d_w)))
    return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  const double v2 = (double)vcore * (double)vcore;
  if (d_xt == d_xs && d_zt == d_zs && nt == ns && use_symmetric(c, (long long)ns, v2 * v2)) {
    long long nt_pad = 0;
    SymOperands o{};
    o.x = d_xs; o.z = d_zs; o.g = d_gs;
    const long long *acc = nullptr, *bad = nullptr;
    CHK(launch_sym(c, o, (long long)ns, v2 * v2, &nt_pad, &acc, &bad));
    hipLaunchKernelGGL(finish_sym, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream, acc, acc + nt_pad, ctx_scale(c),
                       bad, (long long)nt, d_u, d_w);
    HIPCHK(c, hipGetLastError());
    return LUDVM_OK;
  }
  PairArgs a{};
  a.xs = d_xs; a.zs = d_zs; a.gs = d_gs; a.ns = (long long)ns;
  a.xt = d_xt; a.zt = d_zt; a.nt = (long long)nt;
  a.vc4 = v2 * v2;
  return induce_device(c, a, (long long)nt, (long long)ns, LUDVM_PREC_F32, d_u, d_w);
}

int ludvm_advect_dev_f32(ludvm_ctx* c, const float* d_xs, const float* d_zs, const float* d_gs, size_t ns,
                         size_t t_first, size_t nt, float vcore, float dt, float* d_x_out, float* d_z_out) {
  if (!c) return LUDVM_E_ARG;
  if (!d_xs || !d_zs || !d_gs || !d_x_out || !d_z_out) return fail(c, LUDVM_E_ARG, "null array");
  if (t_first + nt > ns) return fail(c, LUDVM_E_ARG, "target range outside the source arrays");
  if (nt == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const double v2 = (double)vcore * (double)vcore;
  if (t_first == 0 && nt == ns && use_symmetric(c, (long long)ns, v2 * v2)) {
    long long nt_pad = 0;
    SymOperands o{};
    o.x = d_xs; o.z = d_zs; o.g = d_gs;
    const long long *acc = nullptr, *bad = nullptr;
    CHK(launch_sym(c, o, (long long)ns, v2 * v2, &nt_pad, &acc, &bad));
    hipLaunchKernelGGL(finish_sym_advect, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream, acc, acc + nt_pad,
                       ctx_scale(c), bad, d_xs, d_zs, 0LL, (long long)nt, dt, d_x_out, d_z_out);
    HIPCHK(c, hipGetLastError());
    return LUDVM_OK;
  }
  PairArgs a{};
  a.xs = d_xs; a.zs = d_zs; a.gs = d_gs; a.ns = (long long)ns;
  a.xt = d_xs + t_first; a.zt = d_zs + t_first; a.nt = (long long)nt;
  a.vc4 = v2 * v2;
  Plan p = make_plan(c, (long long)nt, (long long)ns, LUDVM_PREC_F32);
  CHK(launch_pair(c, a, p, LUDVM_PREC_F32, nullptr, nullptr));  // results stay in the slab
  hipLaunchKernelGGL(finish_advect_f32, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream,
                     static_cast<const float*>(c->part.p), (long long)nt, p.nt_pad, p.nsplit, d_xs, d_zs,
                     (long long)t_first, dt, d_x_out, d_z_out);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

int ludvm_sym_scale_dev_f32(ludvm_ctx* c, const float* d_g, size_t n, float vcore, void* d_scale) {
  if (!c) return LUDVM_E_ARG;
  if (!d_g || !d_scale) return fail(c, LUDVM_E_ARG, "null array");
  const double v2 = (double)vcore * (double)vcore;
  if (!((float)(v2 * v2) > 0.0f)) return fail(c, LUDVM_E_ARG, "the symmetric kernel needs v_core > 0 (fixed-point bound)");
  HIPCHK(c, hipSetDevice(c->device));
  char* rec = static_cast<char*>(d_scale);
  CHK(ensure(c, c->symsc, 128 + (size_t)((n + kPrepChunk - 1) / kPrepChunk) * sizeof(double)));
  return launch_sym_prepare(c, d_g, (long long)n, v2 * v2, reinterpret_cast<SymScale*>(rec), reinterpret_cast<long long*>(rec + 16));
}

int ludvm_sym_accumulate_dev_f32(ludvm_ctx* c, const float* d_x, const float* d_z, const float* d_g, size_t n,
                                 size_t tile_first, size_t tile_count, float vcore, const void* d_scale, long long* d_acc_u,
                                 long long* d_acc_w, long long* d_bad) {
  if (!c) return LUDVM_E_ARG;
  if (!d_x || !d_z || !d_g || !d_scale || !d_acc_u || !d_acc_w || !d_bad) return fail(c, LUDVM_E_ARG, "null array");
  const size_t ntiles = (n + LUDVM_SYM_TILE - 1) / LUDVM_SYM_TILE;
  if (tile_first + tile_count > ntiles) return fail(c, LUDVM_E_ARG, "tile range outside the tile ring");
  if (tile_count == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const double v2 = (double)vcore * (double)vcore;
  SymOperands o{};
  o.x = d_x; o.z = d_z; o.g = d_g;
  o.acc_u = d_acc_u; o.acc_w = d_acc_w;
  o.scale = static_cast<const SymScale*>(d_scale);
  o.bad = d_bad;
  return launch_sym_tiles(c, 8, o, (long long)n, (long long)tile_first, (long long)tile_count, v2 * v2);
}

int ludvm_advect_from_sums_dev_f32(ludvm_ctx* c, const long long* d_sum_u, const long long* d_sum_w, const void* d_scale,
                                   const long long* d_bad, const float* d_x, const float* d_z, size_t t_first, size_t nt,
                                   float dt, float* d_x_out, float* d_z_out) {
  if (!c) return LUDVM_E_ARG;
  if (!d_sum_u || !d_sum_w || !d_scale || !d_bad || !d_x || !d_z || !d_x_out || !d_z_out) return fail(c, LUDVM_E_ARG, "null array");
  if (nt == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(finish_sym_advect, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream, d_sum_u, d_sum_w,
                     static_cast<const SymScale*>(d_scale), d_bad, d_x, d_z, (long long)t_first, (long long)nt, dt, d_x_out,
                     d_z_out);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

}  // extern "C"
