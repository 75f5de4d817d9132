// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: the library's own RCCL communicator (librccl opened at run time: no link dependency) and the all-reduce of a sharded
// roll-up's accumulators.
#include "ctx.hpp"

namespace ludvm_host {

namespace {

// librccl, opened on first use (LUDVM_RCCL_LIB names the file; default librccl.so.1 from the loader's search path -- the
// copy a PyTorch-ROCm process already has mapped, if any).  Single-GPU users never load it.
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl& rccl() {
  // opened once per process, on first use (a function-local static: safe when two contexts' threads get here together)
  static Rccl lib = [] {
    Rccl r;
    const char* name = std::getenv("LUDVM_RCCL_LIB");
    if (!name || !name[0]) name = "librccl.so.1";
    r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) {
      const char* e = dlerror();
      r.error = std::string("cannot open ") + name + ": " + (e ? e : "?");
      return r;
    }
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(r.handle, n);
      if (!p && r.error.empty()) r.error = std::string("librccl has no symbol ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.error.empty()) { dlclose(r.handle); r.handle = nullptr; }
    return r;
  }();
  return lib;
}

int fail_rccl(ludvm_ctx* c, const char* what, ncclResult_t e) {
  Rccl& r = rccl();
  return fail(c, LUDVM_E_COMM, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

#define RCCLCHK(c, call)                                              \
  do {                                                                \
    ncclResult_t e__ = (call);                                        \
    if (e__ != ncclSuccess) return fail_rccl((c), #call, e__);        \
  } while (0)

}  // namespace

int reduce_accumulators(ludvm_ctx* c, long long* acc, long long nt_pad) {
  if (c->shard_world <= 1 && !c->comm_force) return LUDVM_OK;
  if (c->comm) {
    // in place, on the stream the symmetric kernel was launched on and the Euler finisher will be: ONE collective per step
    RCCLCHK(c, rccl().AllReduce(acc - 2, acc - 2, (size_t)(2 * nt_pad + 2), ncclInt64, ncclSum, c->comm, c->stream));
    return LUDVM_OK;
  }
  if (!c->reduce_hook) return fail(c, LUDVM_E_STATE, "sharded roll-up without an all-reduce hook");
  const int rc = c->reduce_hook(c->reduce_user, acc - 2, (size_t)(2 * nt_pad + 2), c->stream);
  if (rc != 0) return fail(c, LUDVM_E_STATE, "the all-reduce hook reported a failure");
  return LUDVM_OK;
}

void comm_release(ludvm_ctx* c) {
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  c->comm = nullptr;
}

}  // namespace ludvm_host

namespace {

// what ludvm_comm_init and ludvm_comm_init_all leave in a context that has joined a communicator
void comm_adopt(ludvm_ctx* c, ncclComm_t comm, int rank, int world, size_t min_vortices) {
  c->comm = comm;
  const char* force = std::getenv("LUDVM_COMM_FORCE");
  c->comm_force = force && force[0] == '1';
  c->comm_rank = rank;
  c->comm_world = world;
  c->shard_rank = rank;
  c->shard_world = world;
  c->shard_min_n = (long long)min_vortices;
  c->reduce_hook = nullptr;
  c->ext_acc = nullptr;
  c->ext_acc_bytes = 0;
}

}  // namespace

extern "C" {

/* ---- the library's own communicator (RCCL over xGMI) --------------------------------------------- */

int ludvm_comm_unique_id(void* id_out, size_t id_bytes) {
  if (!id_out || id_bytes < LUDVM_COMM_ID_BYTES) return LUDVM_E_ARG;
  static_assert(LUDVM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the identifier is RCCL's");
  Rccl& r = rccl();
  if (!r.handle) return LUDVM_E_COMM;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return LUDVM_E_COMM;
  std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return LUDVM_OK;
}

int ludvm_comm_init(ludvm_ctx* c, int rank, int world, const void* id, size_t id_bytes, size_t min_vortices) {
  if (!c) return LUDVM_E_ARG;
  if (world < 1 || rank < 0 || rank >= world) return fail(c, LUDVM_E_ARG, "comm: need 0 <= rank < world");
  if (!id || id_bytes < LUDVM_COMM_ID_BYTES) return fail(c, LUDVM_E_ARG, "comm: the identifier of ludvm_comm_unique_id is 128 bytes");
  if (c->comm) return fail(c, LUDVM_E_STATE, "comm: the context already owns a communicator");
  if (c->shard_world > 1) return fail(c, LUDVM_E_STATE, "comm: the context is sharded through ludvm_set_shard");
  Rccl& r = rccl();
  if (!r.handle) return fail(c, LUDVM_E_COMM, "comm: " + r.error);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  RCCLCHK(c, r.CommInitRank(&comm, world, uid, rank));     // collective: returns when every rank has joined
  comm_adopt(c, comm, rank, world, min_vortices);
  return LUDVM_OK;
}

int ludvm_comm_init_all(ludvm_ctx** ctxs, int n, size_t min_vortices) {
  if (!ctxs || n < 1 || !ctxs[0]) return LUDVM_E_ARG;
  ludvm_ctx* c0 = ctxs[0];                       // (errors are reported on the first context)
  if (n > 64) return fail(c0, LUDVM_E_ARG, "comm: at most 64 contexts");
  int devs[64];
  for (int k = 0; k < n; ++k) {
    ludvm_ctx* c = ctxs[k];
    if (!c) return fail(c0, LUDVM_E_ARG, "comm: null context");
    if (c->comm) return fail(c0, LUDVM_E_STATE, "comm: a context already owns a communicator");
    if (c->shard_world > 1) return fail(c0, LUDVM_E_STATE, "comm: a context is sharded through ludvm_set_shard");
    for (int q = 0; q < k; ++q)
      if (ctxs[q] == c || ctxs[q]->device == c->device)
        return fail(c0, LUDVM_E_ARG, "comm: one context per device, one device per context (RCCL ranks cannot share a GPU)");
    devs[k] = c->device;
  }
  Rccl& r = rccl();
  if (!r.handle) return fail(c0, LUDVM_E_COMM, "comm: " + r.error);
  for (int k = 0; k < n; ++k) {
    HIPCHK(c0, hipSetDevice(ctxs[k]->device));
    HIPCHK(c0, hipStreamSynchronize(ctxs[k]->stream));
  }
  ncclComm_t comms[64];
  RCCLCHK(c0, r.CommInitAll(comms, n, devs));    // one call, this thread: no identifier, nothing to exchange
  for (int k = 0; k < n; ++k) comm_adopt(ctxs[k], comms[k], k, n, min_vortices);
  return LUDVM_OK;
}

int ludvm_comm_destroy(ludvm_ctx* c) {
  if (!c) return LUDVM_E_ARG;
  if (!c->comm) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->stream_b) HIPCHK(c, hipStreamSynchronize(c->stream_b));
  ncclComm_t comm = c->comm;
  c->comm = nullptr;
  c->comm_force = false;
  c->comm_rank = 0; c->comm_world = 1;
  c->shard_rank = 0; c->shard_world = 1; c->shard_min_n = 0;
  RCCLCHK(c, rccl().CommDestroy(comm));
  return LUDVM_OK;
}

int ludvm_comm_info(ludvm_ctx* c, int* rank, int* world) {
  if (!c) return LUDVM_E_ARG;
  if (rank) *rank = c->comm_rank;
  if (world) *world = c->comm ? c->comm_world : 0;
  return LUDVM_OK;
}

int ludvm_comm_allreduce_i64_dev(ludvm_ctx* c, long long* d_buf, size_t count) {
  if (!c) return LUDVM_E_ARG;
  if (!c->comm) return fail(c, LUDVM_E_STATE, "comm: no communicator (ludvm_comm_init)");
  if (count == 0) return LUDVM_OK;
  if (!d_buf) return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  RCCLCHK(c, rccl().AllReduce(d_buf, d_buf, count, ncclInt64, ncclSum, c->comm, c->stream));
  return LUDVM_OK;
}

int ludvm_comm_allgather_dev(ludvm_ctx* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  if (!c) return LUDVM_E_ARG;
  if (!c->comm) return fail(c, LUDVM_E_STATE, "comm: no communicator (ludvm_comm_init)");
  if (bytes_per_rank == 0) return LUDVM_OK;
  if (!d_send || !d_recv) return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  RCCLCHK(c, rccl().AllGather(d_send, d_recv, bytes_per_rank, ncclInt8, c->comm, c->stream));
  return LUDVM_OK;
}

int ludvm_comm_allgather_host(ludvm_ctx* c, const void* send, void* recv, size_t bytes_per_rank) {
  if (!c) return LUDVM_E_ARG;
  if (!c->comm) return fail(c, LUDVM_E_STATE, "comm: no communicator (ludvm_comm_init)");
  if (bytes_per_rank == 0) return LUDVM_OK;
  if (!send || !recv) return fail(c, LUDVM_E_ARG, "null array");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t world = (size_t)c->comm_world;
  CHK(ensure(c, c->arena, Arena::need(bytes_per_rank, 1) + Arena::need(world * bytes_per_rank, 1)));
  Arena ar(c->arena.p);
  char* ds = ar.take<char>(bytes_per_rank);
  char* dr = ar.take<char>(world * bytes_per_rank);
  HIPCHK(c, hipMemcpyAsync(ds, send, bytes_per_rank, hipMemcpyHostToDevice, c->stream));
  RCCLCHK(c, rccl().AllGather(ds, dr, bytes_per_rank, ncclInt8, c->comm, c->stream));
  HIPCHK(c, hipMemcpyAsync(recv, dr, world * bytes_per_rank, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

}  // extern "C"
