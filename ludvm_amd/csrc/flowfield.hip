// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: flow-field grids (LUDVM.flowfield, LUDVM.py:1186-1298) and the vorticity stencil.
#include "ctx.hpp"
#include "field_kernels.hpp"

extern "C" {

/* ---- flow field ---------------------------------------------------------------------------- */

int ludvm_flowfield_dev_f32(ludvm_ctx* c, double xmin, double zmin, double dr, size_t nx, size_t nz, const float* d_xs,
                            const float* d_zs, const float* d_gs, size_t ns, float vcore, float* d_u, float* d_w) {
  if (!c) return LUDVM_E_ARG;
  if (nx == 0 || nz == 0) return LUDVM_OK;
  if ((ns && (!d_xs || !d_zs || !d_gs)) || !d_u || !d_w) return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  PairArgs a{};
  a.xs = d_xs; a.zs = d_zs; a.gs = d_gs; a.ns = (long long)ns;
  a.nt = (long long)(nx * nz);
  a.grid_nz = (long long)nz;
  a.xmin = xmin; a.zmin = zmin; a.dr = dr;
  const double v2 = (double)vcore * (double)vcore;
  a.vc4 = v2 * v2;
  return induce_device(c, a, a.nt, a.ns, LUDVM_PREC_F32, d_u, d_w);
}

// Sources of a flow field from host float64 arrays: uploaded and converted to local-origin fp32 (offsets from the
// origin of each 256-source block; the wake arrives in shedding order, so a block is compact).  The grid targets are
// generated in float64 and referred to each source block's origin, so a flow field over a wake at |x| ~ 50 with
// vortices 1e-3 apart keeps the precision it has near the origin (LUDVM.py:1206, :1216-1217 evaluate in float64).
static int flowfield_upload_local(ludvm_ctx* c, Arena& ar, const double* xs, const double* zs, const double* gs, size_t ns,
                                  PairArgs& a, double* mean_extent, bool* reordered) {
  const size_t nsb = (size_t)origin_slots((long long)ns);
  double* dxs = ar.take<double>(ns);
  double* dzs = ar.take<double>(ns);
  double* dgs = ar.take<double>(ns);
  float* fxs = ar.take<float>(ns);
  float* fzs = ar.take<float>(ns);
  float* fgs = ar.take<float>(ns);
  float* sox = ar.take<float>(nsb);
  float* soz = ar.take<float>(nsb);
  double* oxs = ar.take<double>(ns);
  double* ozs = ar.take<double>(ns);
  double* ogs = ar.take<double>(ns);
  HIPCHK(c, hipMemcpyAsync(dxs, xs, ns * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dzs, zs, ns * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dgs, gs, ns * 8, hipMemcpyHostToDevice, c->stream));
  const dim3 bs(kBlock), gs_(blocks_for((long long)ns));
  OrderWs ow{};
  CHK(order_workspace(c, ns, &ow));
  *mean_extent = 0.0;
  *reordered = false;
  if (ns >= kOrderMin) {
    // sources that are not in a compact order (a turbulence cloud rather than a shed wake) are taken in Morton order: the
    // sum over the sources does not care, the origin classes become compact (every rank of a sharded flow field holds the
    // same sources and derives the same order: the row blocks stay bit for bit the one-GPU rows)
    const unsigned* ord = nullptr;
    CHK(spatial_order_if_needed(c, ow, 0, dxs, dzs, ns, &ord, mean_extent));
    if (ord) {
      CHK(order_gather(c, dxs, dzs, dgs, ord, (long long)ns, oxs, ozs, ogs));
      dxs = oxs; dzs = ozs; dgs = ogs;
      *reordered = true;
    }
  } else {
    // (too few to order; how wide their classes are still decides whether fp32 offsets resolve the core)
    double e = 0.0;
    CHK(class_extent_sum(c, ow, dxs, dzs, nullptr, ns, &e));
    *mean_extent = e / (2.0 * std::ceil((double)ns / kOriginBlock));
  }
  hipLaunchKernelGGL(cvt_f64_to_local, gs_, bs, 0, c->stream, dxs, fxs, sox, (long long)ns);
  hipLaunchKernelGGL(cvt_f64_to_local, gs_, bs, 0, c->stream, dzs, fzs, soz, (long long)ns);
  hipLaunchKernelGGL(cvt_f64_to_f32, gs_, bs, 0, c->stream, dgs, fgs, (float*)nullptr, (long long)ns);
  HIPCHK(c, hipGetLastError());
  a.xs = fxs; a.zs = fzs; a.gs = fgs; a.ns = (long long)ns;
  a.scx = sox; a.scz = soz;
  return LUDVM_OK;
}
static size_t flowfield_upload_bytes(size_t ns) {
  return 6 * Arena::need(ns, 8) + 3 * Arena::need(ns, 4) + 2 * Arena::need((size_t)origin_slots((long long)ns), 4);
}

int ludvm_flowfield_f32(ludvm_ctx* c, double xmin, double zmin, double dr, size_t nx, size_t nz, const double* xs,
                        const double* zs, const double* gs, size_t ns, double vcore, float* u, float* w) {
  return ludvm_flowfield_rows_f32(c, xmin, zmin, dr, nx, nz, 0, nx, xs, zs, gs, ns, vcore, u, w, nullptr);
}

int ludvm_flowfield_vorticity_f32(ludvm_ctx* c, double xmin, double zmin, double dr, size_t nx, size_t nz, const double* xs,
                                  const double* zs, const double* gs, size_t ns, double vcore, float* u, float* w, float* ome) {
  return ludvm_flowfield_rows_f32(c, xmin, zmin, dr, nx, nz, 0, nx, xs, zs, gs, ns, vcore, u, w, ome);
}

}  // extern "C"

// Rows [row_first, row_first + row_count) of the grid, in fp32 on local-origin sources (T = float) or in float64
// throughout (T = double), with the vorticity stencil on the device.
template <typename T>
static int flowfield_rows(ludvm_ctx* c, double xmin, double zmin, double dr, size_t nx, size_t nz, size_t row_first,
                          size_t row_count, const double* xs, const double* zs, const double* gs, size_t ns, double vcore,
                          T* u, T* w, T* ome) {
  constexpr bool f64 = sizeof(T) == 8;
  if (!c) return LUDVM_E_ARG;
  if (row_first + row_count > nx) return fail(c, LUDVM_E_ARG, "rows outside the grid");
  if (row_count == 0 || nz == 0) return LUDVM_OK;
  if ((ns && (!xs || !zs || !gs)) || !u || !w) return fail(c, LUDVM_E_ARG, "null array");
  if (ome && (nx < 2 || nz < 2)) return fail(c, LUDVM_E_ARG, "vorticity needs nx, nz >= 2");
  HIPCHK(c, hipSetDevice(c->device));
  // with the vorticity wanted, one halo row on each interior side: the centred differences of the block's edge rows
  // then need nothing from the rows' other owners (LUDVM.py:1224-1292 keeps one-sided forms for the grid's own edges)
  const size_t h0 = ome && row_first > 0 ? row_first - 1 : row_first;
  const size_t h1 = ome && row_first + row_count < nx ? row_first + row_count + 1 : row_first + row_count;
  const size_t rows = h1 - h0, nt = rows * nz;
  CHK(ensure(c, c->arena, (f64 ? 3 * Arena::need(ns, 8) : flowfield_upload_bytes(ns)) + 3 * Arena::need(nt, sizeof(T))));
  Arena ar(c->arena.p);
  T* du = ar.take<T>(nt);
  T* dw = ar.take<T>(nt);
  T* dome = ar.take<T>(nt);
  if (ns == 0) {
    HIPCHK(c, hipMemsetAsync(du, 0, nt * sizeof(T), c->stream));
    HIPCHK(c, hipMemsetAsync(dw, 0, nt * sizeof(T), c->stream));
  } else {
    PairArgs a{};
    if (f64) {
      double* dxs = ar.take<double>(ns);
      double* dzs = ar.take<double>(ns);
      double* dgs = ar.take<double>(ns);
      HIPCHK(c, hipMemcpyAsync(dxs, xs, ns * 8, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(dzs, zs, ns * 8, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(dgs, gs, ns * 8, hipMemcpyHostToDevice, c->stream));
      a.xs = dxs; a.zs = dzs; a.gs = dgs; a.ns = (long long)ns;
    } else {
      double mean_extent = 0.0;
      bool reordered = false;
      CHK(flowfield_upload_local(c, ar, xs, zs, gs, ns, a, &mean_extent, &reordered));
      if (too_sparse(mean_extent, reordered, vcore)) {
        // Sources too sparse for their core (too_sparse): fp32 offsets cannot resolve it in any order and the grid
        // kernels have no hi+lo variant -- the rows are evaluated in float64 (LUDVM.py:1206, :1216-1217 do) and returned as
        // float32.  Rare: a shed wake sits at 230 v_core, config 5's cloud at 2.
        const size_t cnt = row_count * nz;
        std::vector<double> hu(cnt), hw(cnt), ho(ome ? cnt : 0);
        CHK(flowfield_rows<double>(c, xmin, zmin, dr, nx, nz, row_first, row_count, xs, zs, gs, ns, vcore, hu.data(), hw.data(),
                                   ome ? ho.data() : nullptr));
        for (size_t k = 0; k < cnt; ++k) { u[k] = (T)hu[k]; w[k] = (T)hw[k]; }
        if (ome) for (size_t k = 0; k < cnt; ++k) ome[k] = (T)ho[k];
        return LUDVM_OK;
      }
    }
    a.nt = (long long)nt;
    a.grid_nz = (long long)nz;
    a.grid_row0 = (long long)h0;
    a.xmin = xmin; a.zmin = zmin; a.dr = dr;
    const double v2 = vcore * vcore;
    a.vc4 = v2 * v2;
    // (planned for the whole grid: a block of rows is then bit for bit what the whole-grid call computes for them)
    CHK(induce_device(c, a, a.nt, a.ns, f64 ? LUDVM_PREC_F64 : LUDVM_PREC_F32, du, dw, (long long)(nx * nz)));
  }
  // velocity and vorticity leave the device together: the stencil (LUDVM.py:1224-1292) runs on the fields where they are
  if (ome) {
    if constexpr (f64) {
      hipLaunchKernelGGL(vorticity_f64, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream, du, dw, (long long)rows,
                         (long long)nz, (long long)h0, xmin, zmin, dr, dome);
      HIPCHK(c, hipGetLastError());
    } else {
      CHK(ludvm_vorticity_dev_f32(c, du, dw, rows, nz, (float)dr, dome));
    }
  }
  const size_t off = (row_first - h0) * nz, cnt = row_count * nz;
  HIPCHK(c, hipMemcpyAsync(u, du + off, cnt * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(w, dw + off, cnt * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  if (ome) HIPCHK(c, hipMemcpyAsync(ome, dome + off, cnt * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

extern "C" {

int ludvm_flowfield_rows_f32(ludvm_ctx* c, double xmin, double zmin, double dr, size_t nx, size_t nz, size_t row_first,
                             size_t row_count, const double* xs, const double* zs, const double* gs, size_t ns, double vcore,
                             float* u, float* w, float* ome) {
  return flowfield_rows<float>(c, xmin, zmin, dr, nx, nz, row_first, row_count, xs, zs, gs, ns, vcore, u, w, ome);
}

int ludvm_flowfield_rows_f64(ludvm_ctx* c, double xmin, double zmin, double dr, size_t nx, size_t nz, size_t row_first,
                             size_t row_count, const double* xs, const double* zs, const double* gs, size_t ns, double vcore,
                             double* u, double* w, double* ome) {
  return flowfield_rows<double>(c, xmin, zmin, dr, nx, nz, row_first, row_count, xs, zs, gs, ns, vcore, u, w, ome);
}

int ludvm_vorticity_dev_f32(ludvm_ctx* c, const float* d_u, const float* d_w, size_t nx, size_t nz, float dr,
                            float* d_ome) {
  if (!c) return LUDVM_E_ARG;
  if (nx < 2 || nz < 2) return fail(c, LUDVM_E_ARG, "vorticity needs nx, nz >= 2");
  if (!d_u || !d_w || !d_ome) return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(vorticity_f32, dim3(blocks_for((long long)(nx * nz))), dim3(kBlock), 0, c->stream, d_u, d_w,
                     (long long)nx, (long long)nz, dr, d_ome);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

int ludvm_vorticity_f32(ludvm_ctx* c, const float* u, const float* w, size_t nx, size_t nz, double dr, float* ome) {
  if (!c) return LUDVM_E_ARG;
  if (nx < 2 || nz < 2) return fail(c, LUDVM_E_ARG, "vorticity needs nx, nz >= 2");
  if (!u || !w || !ome) return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nt = nx * nz;
  CHK(ensure(c, c->arena, 3 * Arena::need(nt, 4)));
  Arena ar(c->arena.p);
  float* du = ar.take<float>(nt);
  float* dw = ar.take<float>(nt);
  float* dome = ar.take<float>(nt);
  HIPCHK(c, hipMemcpyAsync(du, u, nt * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dw, w, nt * 4, hipMemcpyHostToDevice, c->stream));
  CHK(ludvm_vorticity_dev_f32(c, du, dw, nx, nz, (float)dr, dome));
  HIPCHK(c, hipMemcpyAsync(ome, dome, nt * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

}  // extern "C"
