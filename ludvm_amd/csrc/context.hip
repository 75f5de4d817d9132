// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: lifecycle, streams, tuning, the error string, device buffers and staging copies.
#include "ctx.hpp"

namespace ludvm_host {

int fail(ludvm_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

int fail_hip(ludvm_ctx* c, const char* what, hipError_t e) {
  return fail(c, e == hipErrorOutOfMemory ? LUDVM_E_NOMEM : LUDVM_E_HIP,
              std::string(what) + ": " + hipGetErrorString(e));
}

int ensure(ludvm_ctx* c, Buf& b, size_t bytes) {
  if (bytes <= b.cap) return LUDVM_OK;
  // the old contents are never needed across a grow; wait for in-flight users, then replace
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->stream_b) HIPCHK(c, hipStreamSynchronize(c->stream_b));
  if (b.p) HIPCHK(c, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = std::max(bytes, (size_t)1 << 20);
  HIPCHK(c, hipMalloc(&b.p, want));
  b.cap = want;
  return LUDVM_OK;
}

// Host -> device copy for entry points that return without synchronizing: the caller may free or
// overwrite its arrays right after the call, so small uploads go through a context-owned pinned
// ring (a wrap waits for the stream); large ones are copied directly and waited for.
int h2d(ludvm_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return LUDVM_OK;
  if (bytes > kPinBytes / 4) {
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LUDVM_OK;
  }
  if (!c->pin) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->pin), kPinBytes, hipHostMallocDefault));
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (c->pin_off + need > kPinBytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pin_off = 0;
  }
  std::memcpy(c->pin + c->pin_off, src, bytes);
  HIPCHK(c, hipMemcpyAsync(dst, c->pin + c->pin_off, bytes, hipMemcpyHostToDevice, c->stream));
  c->pin_off += need;
  return LUDVM_OK;
}

// Small synchronous device -> host read-back through pinned memory (one DMA, one wait).
int d2h_small_sync(ludvm_ctx* c, const void* dsrc, size_t bytes, void** host_view) {
  if (!c->pin_out) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->pin_out), kPinOutBytes, hipHostMallocDefault));
  HIPCHK(c, hipMemcpyAsync(c->pin_out, dsrc, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *host_view = c->pin_out;
  return LUDVM_OK;
}

}  // namespace ludvm_host

extern "C" {

int ludvm_abi_version(void) { return LUDVM_ABI_VERSION; }

int ludvm_create(int device_ordinal, ludvm_ctx** out) {
  const char* small_env = LUDVM_EXP_ENV("LUDVM_SMALL_TILE_MAX");
  if (!out) return LUDVM_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LUDVM_E_NODEVICE;
  if (device_ordinal < 0 || device_ordinal >= ndev) return LUDVM_E_ARG;
  ludvm_ctx* c = new (std::nothrow) ludvm_ctx();
  if (!c) return LUDVM_E_NOMEM;
  c->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&c->prop, device_ordinal) != hipSuccess) {
    delete c;
    return LUDVM_E_HIP;
  }
  if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
    delete c;
    return LUDVM_E_NODEVICE;
  }
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return LUDVM_E_HIP;
  }
  c->stream = c->own_stream;
  if (const char* gk = LUDVM_EXP_ENV("LUDVM_GRID_KERNEL")) {      // row | patch | patch2 | patch4
    const std::string k(gk);
    c->grid_kernel = k == "row" || k == "1" ? 1 : (k == "patch2" ? 3 : (k == "patch4" ? 4 : 2));
  }
  if (const char* fp = LUDVM_EXP_ENV("LUDVM_FEW_PACKED")) c->few_packed = !(fp[0] == '0');
  if (const char* sq = LUDVM_EXP_ENV("LUDVM_SYM_QUAD")) c->sym_quad = !(sq[0] == '0');
  if (const char* sq = LUDVM_EXP_ENV("LUDVM_SYM_QUAD_MIN_TILES")) c->sym_quad_min_tiles = std::max<long long>(16, std::atoll(sq));
  if (const char* xr = LUDVM_EXP_ENV("LUDVM_XCD_RUN")) c->xcd_run = std::max(0, std::atoi(xr));
  if (const char* ti = LUDVM_EXP_ENV("LUDVM_SYM_TAIL_ITEMS")) c->sym_tail_items = std::max<long long>(0, std::atoll(ti));
  if (const char* mx = LUDVM_EXP_ENV("LUDVM_SYM_MIXED")) {        // 1: mixed granularity at every size; 0: at none (A/B measurements)
    if (mx[0] == '1') c->tune_sym_rsplit = -1;
    if (mx[0] == '0') c->tune_sym_rsplit = -2;
  }
  if (small_env) { c->small_tile_max = std::atoll(small_env); c->small_tile_max_f64 = std::min<long long>(c->small_tile_max, 12000); }
  const char* small64_env = LUDVM_EXP_ENV("LUDVM_SMALL_TILE_MAX_F64");
  if (small64_env) c->small_tile_max_f64 = std::atoll(small64_env);
  *out = c;
  return LUDVM_OK;
}

int ludvm_destroy(ludvm_ctx* c) {
  if (!c) return LUDVM_E_ARG;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
  comm_release(c);   // (before its stream and buffers go)
  for (auto& t : c->pending) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
  for (auto& t : c->pool) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
  void* bufs[] = {c->part.p, c->acc.p, c->symsc.p, c->arena.p, c->orderws.p, c->x64, c->z64, c->g64, c->xh, c->xl, c->zh, c->zl, c->g32,
                  c->xr, c->zr, c->cx, c->cz, c->march_tab.p, c->march_kin.p, c->march_rows.p, c->march_state.p, c->march_hist.p};
  for (void* p : bufs)
    if (p) (void)hipFree(p);
  for (auto& e : c->march_ev)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
  if (c->progress) (void)hipHostFree(c->progress);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->pin_out) (void)hipHostFree(c->pin_out);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return LUDVM_OK;
}

const char* ludvm_last_error(const ludvm_ctx* c) { return c ? c->err.c_str() : "null context"; }

int ludvm_device_info(ludvm_ctx* c, int* cu_count, int* clock_khz, long long* hbm_bytes, char* name, int name_len) {
  if (!c) return LUDVM_E_ARG;
  if (cu_count) *cu_count = c->prop.multiProcessorCount;
  if (clock_khz) *clock_khz = c->prop.clockRate;
  if (hbm_bytes) *hbm_bytes = (long long)c->prop.totalGlobalMem;
  if (name && name_len > 0) {
    std::snprintf(name, (size_t)name_len, "%s (%s)", c->prop.name, c->prop.gcnArchName);
  }
  return LUDVM_OK;
}

int ludvm_set_stream(ludvm_ctx* c, void* hip_stream, int external) {
  if (!c) return LUDVM_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t next = external ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  if (next == c->stream) return LUDVM_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));  // nothing of ours may still be in flight on the old one
  c->pin_off = 0;
  c->stream = next;
  return LUDVM_OK;
}

int ludvm_synchronize(ludvm_ctx* c) {
  if (!c) return LUDVM_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

int ludvm_set_tuning(ludvm_ctx* c, int targets_per_lane, int source_splits) {
  if (!c) return LUDVM_E_ARG;
  if (!(targets_per_lane == 0 || targets_per_lane == 1 || targets_per_lane == 2 || targets_per_lane == 4))
    return fail(c, LUDVM_E_ARG, "targets_per_lane must be 0, 1, 2 or 4");
  if (source_splits < 0 || source_splits > kMaxSplit) return fail(c, LUDVM_E_ARG, "source_splits out of range");
  c->tune_tpl = targets_per_lane;
  c->tune_split = source_splits;
  return LUDVM_OK;
}

int ludvm_set_sym_tuning(ludvm_ctx* c, int vortices_per_lane, int rotation_split) {
  if (!c) return LUDVM_E_ARG;
  if (vortices_per_lane != 0 && vortices_per_lane != 4 && vortices_per_lane != 8)
    return fail(c, LUDVM_E_ARG, "vortices_per_lane must be 0 (heuristic), 4 or 8");
#ifdef LUDVM_EXPERIMENTS
  // measurement build: -1 mixed granularity at every size, -2 at none, -4 the quad variant at every size
  const bool code_ok = rotation_split == -1 || rotation_split == -2 || rotation_split == -4;
#else
  const bool code_ok = false;
#endif
  if (rotation_split != 0 && rotation_split != 1 && rotation_split != 2 && rotation_split != 4 && !code_ok)
    return fail(c, LUDVM_E_ARG, "rotation_split must be 0 (by size), 1, 2 or 4");
  c->tune_sym_t = vortices_per_lane;
  c->tune_sym_rsplit = rotation_split;
  return LUDVM_OK;
}

int ludvm_set_shard(ludvm_ctx* c, int rank, int world, size_t min_vortices, ludvm_allreduce_fn allreduce, void* user,
                    void* d_acc, size_t acc_bytes) {
  if (!c) return LUDVM_E_ARG;
  if (world < 1 || rank < 0 || rank >= world) return fail(c, LUDVM_E_ARG, "shard: need 0 <= rank < world");
  if (world > 1 && (!allreduce || !d_acc || acc_bytes < 64)) return fail(c, LUDVM_E_ARG, "shard: world > 1 needs a hook and an accumulator buffer");
  if (c->comm) return fail(c, LUDVM_E_STATE, "shard: the context owns a communicator (ludvm_comm_destroy first)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->shard_rank = rank;
  c->shard_world = world;
  c->shard_min_n = (long long)min_vortices;
  c->reduce_hook = world > 1 ? allreduce : nullptr;
  c->reduce_user = user;
  c->ext_acc = world > 1 ? d_acc : nullptr;
  c->ext_acc_bytes = world > 1 ? acc_bytes : 0;
  return LUDVM_OK;
}

int ludvm_set_symmetric(ludvm_ctx* c, int mode) {
  if (!c) return LUDVM_E_ARG;
  if (mode < 0) return fail(c, LUDVM_E_ARG, "symmetric mode must be >= 0");
  c->sym_mode = mode;
  return LUDVM_OK;
}

}  // extern "C"
