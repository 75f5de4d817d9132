// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: the spatial order of unordered inputs (class extents, Morton keys + rocPRIM sort through spatial_order.hip).
#include "ctx.hpp"
#include "order_kernels.hpp"

namespace ludvm_host {

// ---- spatial order of unordered inputs ----------------------------------------------------------------
// The fp32 kernels keep 1e-5 of max|u| because positions are offsets from the origin of a COMPACT origin class (256-element
// block x index parity).  A shed wake is compact in its stored order; a caller's array or a turbulence cloud
// (LUDVM.py:98-130) is not: 1.3e-4 / 5e-5 of max|u| for 1e5 / 1e6 vortices uniformly random in a 10 x 4 box at x = -55 with
// v_core = 1.3e-3 [MI355X, profiles/r04_unordered_accuracy.txt].  The reference's float64 sum (:565-569) does not depend on
// the order, so the host-pointer entry points may choose their own: Morton order, when -- and only when -- the given order
// is not already compact, so that a shed wake's bits are what they were.

namespace {

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// workspace for orders of up to nmax points in both slots (sized once per call: a later grow would free an order in use)
int order_workspace(ludvm_ctx* c, size_t nmax, OrderWs* w) {
  const size_t nblk = (nmax + kOriginBlock - 1) / kOriginBlock;
  const size_t tb = spatial_order_temp_bytes(nmax);
  CHK(ensure(c, c->orderws, 2 * up256(nmax * 4) + up256(nblk * 5 * 8) + 256 + tb));
  char* p = static_cast<char*>(c->orderws.p);
  w->order[0] = reinterpret_cast<unsigned*>(p); p += up256(nmax * 4);
  w->order[1] = reinterpret_cast<unsigned*>(p); p += up256(nmax * 4);
  w->ext = reinterpret_cast<double*>(p); p += up256(nblk * 5 * 8);
  w->sum = reinterpret_cast<double*>(p); p += 256;
  w->tmp = p;
  w->tmp_bytes = tb;
  return LUDVM_OK;
}

// Sum of the class extents of (dx, dz)[0, n) taken in `order` (nullptr: as stored), and the set's bounding box
// box[4] = xmin, xmax, zmin, zmax (may be NULL) -> host.  Synchronizes the stream.
int class_extent_sum(ludvm_ctx* c, const OrderWs& w, const double* dx, const double* dz, const unsigned* order, size_t n, double* out,
                     double* box) {
  const long long nblk = (long long)((n + kOriginBlock - 1) / kOriginBlock);
  hipLaunchKernelGGL(class_extents, dim3((unsigned)nblk), dim3(kOriginBlock), 0, c->stream, dx, dz, order, (long long)n, w.ext);
  hipLaunchKernelGGL(reduce_extents, dim3(1), dim3(256), 0, c->stream, w.ext, nblk, w.sum);
  HIPCHK(c, hipGetLastError());
  void* hv = nullptr;
  CHK(d2h_small_sync(c, w.sum, 5 * sizeof(double), &hv));
  const double* r = static_cast<const double*>(hv);
  *out = r[0];
  if (box) for (int k = 0; k < 4; ++k) box[k] = r[1 + k];
  return LUDVM_OK;
}

// Decide whether the n points (device float64 dx, dz) should be taken in Morton
// order, and build that order in slot `slot` of the workspace.  *order_out = the permutation (position k holds the caller's
// element order[k]) or nullptr when the given order stays: fewer than kOrderMin points, classes already as compact as an
// area-filling arrangement would make them (3 x), or not at least 1.5 x less compact than the Morton order makes them.
// *mean_extent = mean over the origin classes of (xmax - xmin) + (zmax - zmin) in the order that was chosen (0 when the
// set was not examined).
int spatial_order_if_needed(ludvm_ctx* c, const OrderWs& w, int slot, const double* dx, const double* dz, size_t n,
                            const unsigned** order_out, double* mean_extent) {
  *order_out = nullptr;
  if (mean_extent) *mean_extent = 0.0;
  if (n < kOrderMin) return LUDVM_OK;
  double e_given = 0.0, box[4];
  CHK(class_extent_sum(c, w, dx, dz, nullptr, n, &e_given, box));
  const double x0 = box[0], z0 = box[2], ex = box[1] - box[0], ez = box[3] - box[2];
  if (!(ex >= 0.0) || !(ez >= 0.0) || (ex == 0.0 && ez == 0.0)) return LUDVM_OK;      // nothing finite, or one point
  const double nclass = 2.0 * std::ceil((double)n / kOriginBlock);
  if (mean_extent) *mean_extent = e_given / nclass;
  const double side = std::sqrt(128.0 * ex * ez / (double)n);        // an area-filling class of 128 points
  if (e_given <= 3.0 * nclass * 2.0 * side) return LUDVM_OK;
  // a thin set (every shed wake: ez << ex makes `side` tiny, the test above never passes): stored along a line from corner
  // to corner of its box a class -- 128 of 256 consecutive points -- spans (ex + ez) 256 / n, and no order packs a line
  // tighter.  Within 3 x of that the given order stays without the keys, the sort and the second pass.
  if (e_given <= 3.0 * nclass * (ex + ez) * 256.0 / (double)n) return LUDVM_OK;
  const double span = std::max(ex, ez);
  OrderBox box_k{x0, z0, 65535.0 / span, 65535.0 / span};
  HIPCHK(c, spatial_order_sort(dx, dz, n, box_k, w.tmp, w.tmp_bytes, w.order[slot], c->stream));
  double e_sorted = 0.0;
  CHK(class_extent_sum(c, w, dx, dz, w.order[slot], n, &e_sorted));
  if (e_given <= 1.5 * e_sorted) return LUDVM_OK;
  *order_out = w.order[slot];
  if (mean_extent) *mean_extent = e_sorted / nclass;
  return LUDVM_OK;
}

int order_gather(ludvm_ctx* c, const double* s0, const double* s1, const double* s2, const unsigned* order, long long n, double* d0,
                 double* d1, double* d2) {
  hipLaunchKernelGGL(gather_f64, dim3(blocks_for(n)), dim3(kBlock), 0, c->stream, s0, s1, s2, order, n, d0, d1, d2);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

int order_scatter(ludvm_ctx* c, const double* s0, const double* s1, const unsigned* order, long long n, double* d0, double* d1) {
  hipLaunchKernelGGL(scatter_f64, dim3(blocks_for(n)), dim3(kBlock), 0, c->stream, s0, s1, order, n, d0, d1);
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

}  // namespace ludvm_host

extern "C" {

int ludvm_spatial_order(ludvm_ctx* c, const double* x, const double* z, size_t n, unsigned* order, int* reordered,
                        double* mean_class_extent) {
  if (!c) return LUDVM_E_ARG;
  if (n && (!x || !z || !order)) return fail(c, LUDVM_E_ARG, "null array");
  if (n >= ((size_t)1 << 32)) return fail(c, LUDVM_E_ARG, "too many points");
  if (reordered) *reordered = 0;
  if (mean_class_extent) *mean_class_extent = 0.0;
  for (size_t i = 0; i < n; ++i) order[i] = (unsigned)i;
  if (n < kOrderMin) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  CHK(ensure(c, c->arena, 2 * Arena::need(n, 8)));
  Arena ar(c->arena.p);
  double* dx = ar.take<double>(n);
  double* dz = ar.take<double>(n);
  HIPCHK(c, hipMemcpyAsync(dx, x, n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dz, z, n * 8, hipMemcpyHostToDevice, c->stream));
  OrderWs ow{};
  CHK(order_workspace(c, n, &ow));
  const unsigned* ord = nullptr;
  CHK(spatial_order_if_needed(c, ow, 0, dx, dz, n, &ord, mean_class_extent));
  if (!ord) return LUDVM_OK;
  HIPCHK(c, hipMemcpyAsync(order, ord, n * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (reordered) *reordered = 1;
  return LUDVM_OK;
}

}  // extern "C"
