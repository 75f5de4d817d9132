// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the
// reference file:line each entry point replaces).  gfx950 only; no CPU path: every entry point
// either runs the HIP kernels or returns an error code.
#include "../../include/ludvm_hip.h"
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"
#include "march_kernels.hpp"
#include "order_kernels.hpp"
#include "spatial_order.hpp"

#include <hip/hip_ext.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl is opened at run time by ludvm_comm_init (no link dependency)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace ludvm;

// Measurement switches.  A PRODUCTION build reads two environment variables, neither of which can change a result bit:
// LUDVM_RCCL_LIB (which librccl to open) and LUDVM_COMM_FORCE (a one-rank communicator issues its collectives: tests).
// Everything else -- the A/B switches of rounds 1-3 (kernel variants, thresholds, chunking; six of them change the
// partition into fp32 partial sums and hence result bits) and the negative codes of ludvm_set_sym_tuning -- exists only in
// the measurement build, `make libludvm_hip_exp.so` (-DLUDVM_EXPERIMENTS), which tools/ and the tests of forced variants
// load.  In a production build the names below do not even reach the object file (tests/test_cabi.py checks).
#ifdef LUDVM_EXPERIMENTS
#define LUDVM_EXP_ENV(name) std::getenv(name)
#else
#define LUDVM_EXP_ENV(name) static_cast<const char*>(nullptr)
#endif

namespace {

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};

struct TimedLaunch {
  hipEvent_t e0, e1;
};

}  // namespace

struct ludvm_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop;
  std::string err;

  int tune_tpl = 0;
  int tune_split = 0;
  int sym_mode = 1;
  int tune_sym_t = 0, tune_sym_rsplit = 0;   // ludvm_set_sym_tuning (0 = heuristics)
  int xcd_run = 0;                           // chunks per run of the XCD placement (LUDVM_XCD_RUN; 0 = a launch's chunks / 8)
  long long sym_tail_items = kSymTailItems;  // mixed granularity: work kept for the fine-grained end (LUDVM_SYM_TAIL_ITEMS)
  int grid_kernel = 2;                       // flow-field grids (LUDVM_GRID_KERNEL): 1 = 4 points of a row per lane; 2 = patch,
                                             // 4 x 4 from 2^20 grid points and 2 x 4 below; 3 / 4 = always the 2 x 4 / 4 x 4 patch
  long long small_tile_max = 14000;          // direct fp32 launches with at most this many sources use 256-source tiles
  bool sym_quad = true;                      // large symmetric launches: four I tiles per workgroup share each partner tile (LUDVM_SYM_QUAD=0: off)
  long long sym_quad_min_tiles = 640;        //   ... from this many 512-vortex tiles on (LUDVM_SYM_QUAD_MIN_TILES)
  bool few_packed = true;                    // fp64 launches with <= 128 targets: several source splits per workgroup (LUDVM_FEW_PACKED=0: off)
  long long small_tile_max_f64 = 12000;      // fp64 launches with at most this many sources use 128-source tiles
                                             // (roll-up step 52 -> 26 us at 2400 vortices, 87 -> 72 at 8192 [MI355X])

  Buf part;   // partial slabs of the split reduction
  Buf acc;    // raw (u, w) sums of the symmetric kernel: [2][nt_pad] 64-bit fixed-point integers
  Buf symsc;  // SymScale of the current symmetric launch, followed by its NaN counter (long long)
  Buf arena;  // staging for the host-pointer entry points
  Buf orderws;  // spatial order of unordered inputs: two permutations, class extents, the sort's temporaries
  char* pin = nullptr;  // pinned host ring for small uploads from entry points that do not synchronize
  size_t pin_off = 0;
  char* pin_out = nullptr;  // pinned host buffer for small synchronous read-backs
  std::vector<double> pack;  // host staging of a time step's packed upload

  // resident wake (float64 master + fp32 mirrors)
  size_t wake_cap = 0, wake_n = 0;
  double *x64 = nullptr, *z64 = nullptr, *g64 = nullptr;
  float *xh = nullptr, *xl = nullptr, *zh = nullptr, *zl = nullptr, *g32 = nullptr;
  float *xr = nullptr, *zr = nullptr, *cx = nullptr, *cz = nullptr;   // local-origin offsets and block origins
  Mirrors mir() const { return Mirrors{xh, xl, zh, zl, xr, zr, cx, cz}; }

  // device-resident march (ludvm_march_setup / ludvm_march_run)
  Buf march_tab, march_kin, march_rows, march_state, march_hist;
  MarchSetup msetup{};
  size_t march_kin_rows = 0;
  double march_vcore = 0.0;
  bool march_ready = false;
  unsigned long long* progress = nullptr;      // host-mapped ring: slot s % kProgressRing = (s << 32 | wake size after step s)
  unsigned long long* progress_dev = nullptr;
  hipEvent_t march_ev[2] = {nullptr, nullptr};
  hipStream_t stream_b = nullptr;              // the solve chain beside the roll-up (overlapped march steps)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;

  // sharded roll-up (ludvm_set_shard): this context evaluates tile block `shard_rank` of `shard_world`; the hook sums
  // the fixed-point accumulators over the contexts / processes before every Euler finisher
  int shard_rank = 0, shard_world = 1;
  long long shard_min_n = 0;    // wakes smaller than this are not worth a collective per step: every owner does them whole
  ludvm_allreduce_fn reduce_hook = nullptr;
  void* reduce_user = nullptr;
  void* ext_acc = nullptr;      // caller-owned accumulator memory (e.g. a torch tensor the hook all-reduces)
  size_t ext_acc_bytes = 0;
  // the library's own RCCL communicator (ludvm_comm_init): the all-reduce is then issued here, on the context's stream
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  bool comm_force = false;      // LUDVM_COMM_FORCE=1: issue the collectives even in a one-rank communicator (tests)

  // kernel timing
  bool timing = false;
  std::vector<TimedLaunch> pending;
  std::vector<TimedLaunch> pool;
  double total_ms = 0.0;
  long long launches = 0;
};

namespace {

int fail(ludvm_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

int fail_hip(ludvm_ctx* c, const char* what, hipError_t e) {
  return fail(c, e == hipErrorOutOfMemory ? LUDVM_E_NOMEM : LUDVM_E_HIP,
              std::string(what) + ": " + hipGetErrorString(e));
}

#define HIPCHK(c, call)                                     \
  do {                                                      \
    hipError_t e__ = (call);                                \
    if (e__ != hipSuccess) return fail_hip((c), #call, e__); \
  } while (0)

#define CHK(call)                     \
  do {                                \
    int rc__ = (call);                \
    if (rc__ != LUDVM_OK) return rc__; \
  } while (0)

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

constexpr size_t kPinBytes = (size_t)1 << 20;

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

constexpr size_t kPinOutBytes = (size_t)1 << 16;

// Small synchronous device -> host read-back through pinned memory (one DMA, one wait).
int d2h_small_sync(ludvm_ctx* c, const void* dsrc, size_t bytes, void** host_view) {
  if (!c->pin_out) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->pin_out), kPinOutBytes, hipHostMallocDefault));
  HIPCHK(c, hipMemcpyAsync(c->pin_out, dsrc, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *host_view = c->pin_out;
  return LUDVM_OK;
}

// bump allocator over the staging arena
struct Arena {
  char* base;
  size_t off = 0;
  explicit Arena(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    T* r = reinterpret_cast<T*>(base + off);
    off += (n * sizeof(T) + 255) & ~(size_t)255;
    return r;
  }
  static size_t need(size_t n, size_t elt) { return (n * elt + 255) & ~(size_t)255; }
};

inline unsigned blocks_for(long long n) { return (unsigned)((n + kBlock - 1) / kBlock); }

struct Plan {
  int tpl;
  int tile;
  int nsplit;
  long long chunk;
  long long nt_pad;
  dim3 grid;
};

constexpr int kTileF32 = 1024;
// Small source sets (a young wake, a chord-sized launch): with 1024-source tiles a launch of a few hundred sources is
// one tile walked by one wave per SIMD, which issues at ~40 % of the SIMD's rate; 256-source tiles give 4x more splits
// (more workgroups, shorter walks): one self-advection step 21 -> 9-11 us up to 4096 vortices, 46 -> 37 us at
// 12 288, no gain at 16 384 [MI355X].  LUDVM_SMALL_TILE_MAX (sources) overrides the switch-over, 0 disables.
constexpr int kTileF32Small = 256;
constexpr int kTileF64 = 512;
constexpr int kTileF64Few = 128;         // fp64 launches with few targets (chord points): short tiles, more workgroups
constexpr long long kFewTargets = 256;
constexpr long long kTargetBlocks = 16384;  // total workgroups aimed for (2048 resident at 8/CU)
constexpr int kMaxSplit = 2048;

// small_ok = false: the launch has no 256-source-tile kernel (generic flow-field grids), keep the chunk a multiple of 1024
// plan_nt: the target count that DECIDES the plan -- tile size, targets per lane and the split of the sources into
// partial sums, i.e. everything the rounding of a result depends on -- when the launch itself covers only a part of a
// larger target set (a block of rows of a flow-field grid: the block then carries the whole grid's bits); 0 = nt.
Plan make_plan(const ludvm_ctx* c, long long nt_launch, long long ns, int precision, bool small_ok = true, long long plan_nt = 0,
               bool grid_patch = false) {
  Plan p{};
  const long long nt = plan_nt > 0 ? plan_nt : nt_launch;
  const bool f64 = precision == LUDVM_PREC_F64;
  p.tile = f64 ? ((nt <= kFewTargets || ns <= c->small_tile_max_f64) ? kTileF64Few : kTileF64) : kTileF32;
  if (!f64 && small_ok && ns <= c->small_tile_max) p.tile = kTileF32Small;
  if (f64) {
    p.tpl = 1;
  } else if (c->tune_tpl == 1 || c->tune_tpl == 2 || c->tune_tpl == 4) {
    p.tpl = c->tune_tpl;
  } else {
    p.tpl = nt >= 131072 ? 2 : 1;
  }
  // the small tile exists for TPL = 1 (and for the 4-points-per-lane grid kernel, whose TPL the launch fixes itself)
  if (p.tile == kTileF32Small && (p.tpl != 1 || nt > 65536)) p.tile = kTileF32;
  // target tiles = workgroups per source split.  The flow-field patch kernels hold 8 or 16 grid points per lane, not tpl:
  // counted with tpl, a 4096 x 4096 grid looked like 32 768 workgroups and got ONE source split -- 4096 workgroups that
  // each walk all the sources for a quarter of a second, and a launch that ends over half such a lifetime (config 5:
  // 8.05e12 pairs/s with one split, 8.13e12 with four, 8.15e12 with eight [MI355X])
  const long long per_wg = grid_patch ? (long long)kBlock * 4 * (nt >= (1LL << 20) ? 4 : 2) : (long long)kBlock * p.tpl;
  const long long ttiles = std::max<long long>(1, (nt + per_wg - 1) / per_wg);
  const long long max_split = std::max<long long>(1, (ns + p.tile - 1) / p.tile);
  long long nsplit = c->tune_split > 0 ? c->tune_split : (kTargetBlocks + ttiles - 1) / ttiles;
  nsplit = std::max<long long>(1, std::min<long long>(std::min<long long>(nsplit, max_split), kMaxSplit));
  long long chunk = (std::max<long long>(ns, 1) + nsplit - 1) / nsplit;
  chunk = (chunk + p.tile - 1) / p.tile * p.tile;
  p.chunk = chunk;
  p.nsplit = (int)std::max<long long>(1, (ns + chunk - 1) / chunk);
  p.nt_pad = (nt_launch + 63) / 64 * 64;
  const long long tiles_launch = std::max<long long>(1, (nt_launch + (long long)kBlock * p.tpl - 1) / ((long long)kBlock * p.tpl));
  p.grid = dim3((unsigned)tiles_launch, (unsigned)p.nsplit, 1);
  return p;
}

int timed_begin(ludvm_ctx* c, TimedLaunch& t, bool& active) {
  active = c->timing;
  if (!active) return LUDVM_OK;
  if (!c->pool.empty()) {
    t = c->pool.back();
    c->pool.pop_back();
  } else {
    HIPCHK(c, hipEventCreate(&t.e0));
    HIPCHK(c, hipEventCreate(&t.e1));
  }
  HIPCHK(c, hipEventRecord(t.e0, c->stream));
  return LUDVM_OK;
}

int timed_end(ludvm_ctx* c, TimedLaunch& t, bool active) {
  if (!active) return LUDVM_OK;
  HIPCHK(c, hipEventRecord(t.e1, c->stream));
  c->pending.push_back(t);
  return LUDVM_OK;
}

int drain_timing(ludvm_ctx* c) {
  for (auto& t : c->pending) {
    HIPCHK(c, hipEventSynchronize(t.e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, t.e0, t.e1));
    c->total_ms += ms;
    c->launches += 1;
    c->pool.push_back(t);
  }
  c->pending.clear();
  return LUDVM_OK;
}

// workgroups of a flow-field grid launch: 256 lanes of 4 row points (patch_rows = 0), or of patch_rows x 4 patches
long long grid_kernel_blocks(const PairArgs& a, int patch_rows) {
  if (patch_rows > 0) {
    const long long nrows = a.nt / a.grid_nz, patches = ((nrows + patch_rows - 1) / patch_rows) * (a.grid_nz / 4);
    return (patches + kBlock - 1) / kBlock;
  }
  return (a.nt + (long long)kBlock * 4 - 1) / ((long long)kBlock * 4);
}

// rows of the patch of grid points a lane owns (4 columns), 0 = the row kernel.  Every variant performs the same
// operations on the same operands for a given grid point, so the choice never changes a result bit.
constexpr long long kPatch4MinTargets = 1LL << 20;
int grid_patch_rows(const ludvm_ctx* c, const PairArgs& a, const Plan& p) {
  if (c->grid_kernel == 1) return 0;
  if (c->grid_kernel == 3 || p.tile == kTileF32Small) return 2;       // (the 4 x 4 patch exists for 1024-source tiles)
  if (c->grid_kernel == 4) return 4;
  return a.nt >= kPatch4MinTargets ? 4 : 2;
}

// Launch the main pair kernel described by `a` (sources, targets and vc4 filled in by the caller)
// under plan `p`; a.part / a.u / a.w / a.nt_pad / a.chunk are completed here.  With more than one
// split the results are left in c->part for a finisher; with one split they go to (u, w).
int launch_pair(ludvm_ctx* c, PairArgs a, const Plan& p, int precision, void* u, void* w) {
  const size_t elt = precision == LUDVM_PREC_F64 ? sizeof(double) : sizeof(float);
  a.chunk = p.chunk;
  a.nt_pad = p.nt_pad;
  a.nsplit = p.nsplit;
  a.u = u;
  a.w = w;
  a.part = nullptr;
  if (p.nsplit > 1 || u == nullptr) {
    CHK(ensure(c, c->part, (size_t)p.nsplit * 2 * (size_t)p.nt_pad * elt));
    a.part = c->part.p;
  }
  dim3 grid = p.grid;
  // a single split asked to land in the slab (fused finisher): the kernel distinguishes by
  // gridDim.y, so give it the direct pointers into slab row 0
  if (p.nsplit == 1 && u == nullptr) {
    a.u = c->part.p;
    a.w = static_cast<char*>(c->part.p) + (size_t)p.nt_pad * elt;
  }
  TimedLaunch t{};
  bool active = false;
  CHK(timed_begin(c, t, active));
  if (precision == LUDVM_PREC_F64) {
    // few array targets, results in the slab: several source splits per workgroup (launch sized from the host's bound on
    // the target count when that lives on the device: never, for these launches)
    const long long nt_few = a.nt_dev ? 0 : a.nt;
    if (p.tile == kTileF64Few && c->few_packed && a.part != nullptr && a.grid_nz == 0 && nt_few >= 1 && 2 * nt_few <= kBlock &&
        grid.x == 1 && p.nsplit > 1) {
      const int groups = (int)std::min<long long>(kBlock / nt_few, kFewGroupsMax);
      grid = dim3(1, (unsigned)((p.nsplit + groups - 1) / groups), 1);
      hipLaunchKernelGGL((pair_f64_few<kTileF64Few>), grid, dim3(kBlock), 0, c->stream, a);
    } else if (p.tile == kTileF64Few)
      hipLaunchKernelGGL((pair_f64<kTileF64Few>), grid, dim3(kBlock), 0, c->stream, a);
    else
      hipLaunchKernelGGL((pair_f64<kTileF64>), grid, dim3(kBlock), 0, c->stream, a);
  } else if (a.scx != nullptr) {
    // local-origin fp32 (LUDVM_PREC_F32 wherever the library lays the positions out itself)
    if (a.grid_nz > 0 && a.grid_nz % 4 == 0 && c->tune_tpl == 0) {
      // flow-field grid: a 2 x 4 patch (or 4 points of a row) per lane; the plan's grid is recomputed for it
      const int prows = grid_patch_rows(c, a, p);
      grid = dim3((unsigned)grid_kernel_blocks(a, prows), grid.y, 1);
      if (prows == 4) {
        hipLaunchKernelGGL((pair_f32<16, kTileF32, false, 2, true>), grid, dim3(kBlock), 0, c->stream, a);
      } else if (prows == 2) {
        if (p.tile == kTileF32Small) hipLaunchKernelGGL((pair_f32<8, kTileF32Small, false, 2, true>), grid, dim3(kBlock), 0, c->stream, a);
        else hipLaunchKernelGGL((pair_f32<8, kTileF32, false, 2, true>), grid, dim3(kBlock), 0, c->stream, a);
      } else {
        if (p.tile == kTileF32Small) hipLaunchKernelGGL((pair_f32<4, kTileF32Small, false, 1, true>), grid, dim3(kBlock), 0, c->stream, a);
        else hipLaunchKernelGGL((pair_f32<4, kTileF32, false, 1, true>), grid, dim3(kBlock), 0, c->stream, a);
      }
    } else if (p.tile == kTileF32Small) {
      hipLaunchKernelGGL((pair_f32<1, kTileF32Small, false, 0, true>), grid, dim3(kBlock), 0, c->stream, a);
    } else {
      switch (p.tpl) {
        case 1: hipLaunchKernelGGL((pair_f32<1, kTileF32, false, 0, true>), grid, dim3(kBlock), 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL((pair_f32<2, kTileF32, false, 0, true>), grid, dim3(kBlock), 0, c->stream, a); break;
        default: hipLaunchKernelGGL((pair_f32<4, kTileF32, false, 0, true>), grid, dim3(kBlock), 0, c->stream, a); break;
      }
    }
  } else if (p.tile == kTileF32Small && !(a.grid_nz > 0)) {
    if (precision == LUDVM_PREC_F32X2)
      hipLaunchKernelGGL((pair_f32<1, kTileF32Small, true>), grid, dim3(kBlock), 0, c->stream, a);
    else
      hipLaunchKernelGGL((pair_f32<1, kTileF32Small, false>), grid, dim3(kBlock), 0, c->stream, a);
  } else if (precision == LUDVM_PREC_F32X2) {
    switch (p.tpl) {
      case 1: hipLaunchKernelGGL((pair_f32<1, kTileF32, true>), grid, dim3(kBlock), 0, c->stream, a); break;
      case 2: hipLaunchKernelGGL((pair_f32<2, kTileF32, true>), grid, dim3(kBlock), 0, c->stream, a); break;
      default: hipLaunchKernelGGL((pair_f32<4, kTileF32, true>), grid, dim3(kBlock), 0, c->stream, a); break;
    }
  } else if (a.grid_nz > 0 && a.grid_nz % 4 == 0 && c->tune_tpl == 0) {
    // flow-field grid: a 2 x 4 patch (or 4 points of a row) per lane; the plan's grid is recomputed for it
    const int prows = grid_patch_rows(c, a, p);
    grid = dim3((unsigned)grid_kernel_blocks(a, prows), grid.y, 1);
    if (prows == 4) {
      hipLaunchKernelGGL((pair_f32<16, kTileF32, false, 2>), grid, dim3(kBlock), 0, c->stream, a);
    } else if (prows == 2) {
      if (p.tile == kTileF32Small) hipLaunchKernelGGL((pair_f32<8, kTileF32Small, false, 2>), grid, dim3(kBlock), 0, c->stream, a);
      else hipLaunchKernelGGL((pair_f32<8, kTileF32, false, 2>), grid, dim3(kBlock), 0, c->stream, a);
    } else {
      if (p.tile == kTileF32Small) hipLaunchKernelGGL((pair_f32<4, kTileF32Small, false, 1>), grid, dim3(kBlock), 0, c->stream, a);
      else hipLaunchKernelGGL((pair_f32<4, kTileF32, false, 1>), grid, dim3(kBlock), 0, c->stream, a);
    }
  } else {
    switch (p.tpl) {
      case 1: hipLaunchKernelGGL((pair_f32<1, kTileF32, false>), grid, dim3(kBlock), 0, c->stream, a); break;
      case 2: hipLaunchKernelGGL((pair_f32<2, kTileF32, false>), grid, dim3(kBlock), 0, c->stream, a); break;
      default: hipLaunchKernelGGL((pair_f32<4, kTileF32, false>), grid, dim3(kBlock), 0, c->stream, a); break;
    }
  }
  HIPCHK(c, hipGetLastError());
  CHK(timed_end(c, t, active));
  return LUDVM_OK;
}

// pair kernel + split reduction into (u, w) device arrays of the precision's type
int induce_device(ludvm_ctx* c, const PairArgs& a, long long nt, long long ns, int precision, void* u, void* w,
                  long long plan_nt = 0) {
  if (nt == 0) return LUDVM_OK;
  const size_t elt = precision == LUDVM_PREC_F64 ? sizeof(double) : sizeof(float);
  if (ns == 0) {
    HIPCHK(c, hipMemsetAsync(u, 0, (size_t)nt * elt, c->stream));
    HIPCHK(c, hipMemsetAsync(w, 0, (size_t)nt * elt, c->stream));
    return LUDVM_OK;
  }
  const bool grid_generic = a.grid_nz > 0 && !(a.grid_nz % 4 == 0 && c->tune_tpl == 0);
  // (whatever form of the grid kernel runs -- LUDVM_GRID_KERNEL can force one --: the plan, and with it the bits, stays the same)
  const bool grid_patch = a.grid_nz > 0 && !grid_generic && precision == LUDVM_PREC_F32;
  Plan p = make_plan(c, nt, ns, precision, !grid_generic, plan_nt, grid_patch);
  CHK(launch_pair(c, a, p, precision, u, w));
  if (p.nsplit > 1) {
    if (precision == LUDVM_PREC_F64)
      hipLaunchKernelGGL(finish_sum<double>, dim3(blocks_for(nt)), dim3(kBlock), 0, c->stream,
                         static_cast<const double*>(c->part.p), nt, p.nt_pad, p.nsplit, static_cast<double*>(u),
                         static_cast<double*>(w));
    else
      hipLaunchKernelGGL(finish_sum<float>, dim3(blocks_for(nt)), dim3(kBlock), 0, c->stream,
                         static_cast<const float*>(c->part.p), nt, p.nt_pad, p.nsplit, static_cast<float*>(u),
                         static_cast<float*>(w));
    HIPCHK(c, hipGetLastError());
  }
  return LUDVM_OK;
}

constexpr long long kSymMinN = 16384;   // below this the direct kernel's launch is as fast
// Vortices per lane of the symmetric kernel: 8 (tile 512, 158-162 VGPRs: 3 waves/SIMD) from ~4e4 vortices up, where
// halving the rotation / LDS-read cost per pair wins 2-7 % (with the rotation steps of a tile pair shared by two or four
// waves below ~8e4); 4 (tile 256, 70-90 VGPRs) below, where more and smaller tiles balance better, and for hi+lo
// positions (not instantiated for the 512-vortex tile: hi+lo is instruction-bound either way).
constexpr long long kSymT8MinN = 34816;
static_assert(64 * 8 == LUDVM_SYM_TILE, "the multi-GPU entry points always use the 512-vortex tile");

// The symmetric kernel accumulates in fixed point, which needs the bound sum|Gamma| / (sqrt(2) v_core) on the raw
// sums: point vortices (v_core = 0, or so small that v_core^4 vanishes in fp32) take the direct kernel.
// In the march a symmetric step is an OVERLAPPED step: chord sums and solve run beside the kernel instead of in front
// of it (~40 us of a ~60 us serial step at 1e4 vortices), so it pays earlier there: from ~11 000 vortices [MI355X]
// (profiles/r02_march_symmetric_threshold.txt).
constexpr long long kSymMinNMarch = 11264;
long long sym_threshold(const ludvm_ctx* c, bool march = false) {
  return c->sym_mode == 1 ? (march ? kSymMinNMarch : kSymMinN) : (long long)c->sym_mode;
}
bool use_symmetric(const ludvm_ctx* c, long long n, double vc4, bool march = false) {
  if (c->sym_mode == 0 || !((float)vc4 > 0.0f)) return false;
  return n >= sym_threshold(c, march);
}

int sym_tile_t(const ludvm_ctx* c, long long n, bool hilo, bool local = false) {
  (void)local;                        // local origins fit both tiles (a 512-vortex tile keeps its targets twice)
  if (hilo) return 4;                 // hi+lo positions: 256-vortex tile only
  if (c->tune_sym_t == 4 || c->tune_sym_t == 8) return c->tune_sym_t;
  return n >= kSymT8MinN ? 8 : 4;
}

struct SymOperands {
  const float* x; const float* z; const float* g;
  const float* xl = nullptr; const float* zl = nullptr;     // hi+lo positions (T = 4)
  const float* cx = nullptr; const float* cz = nullptr;     // local origins: x, z are offsets from them
  long long* acc_u; long long* acc_w;
  const SymScale* scale; long long* bad;
};

// Symmetric kernel over I tiles [i_first, i_first + i_count) of the tile ring of (x, z, g)[0, n); raw fixed-point
// sums are ADDED into acc_u / acc_w (n each, zeroed by the caller).  The partition of the work into partial sums
// is a function of n and T alone (sym_geometry).  n_dev (march): the vortex count is read on the device; it lies in
// [n_lo, n], and the grid is sized for the largest wave count any such n needs.
int launch_sym_tiles(ludvm_ctx* c, int T, const SymOperands& o, long long n, long long i_first, long long i_count, double vc4,
                     const long long* n_dev = nullptr, long long n_lo = 0, bool sharded = false) {
  if (n >= (1LL << 31)) return fail(c, LUDVM_E_ARG, "the symmetric kernel indexes vortices with 32 bits: n < 2^31");
  SymArgs a{};
  a.x = o.x; a.z = o.z; a.g = o.g; a.n = n;
  a.n_dev = n_dev;
  a.xl = o.xl; a.zl = o.zl;
  a.cx = o.cx; a.cz = o.cz;
  const bool hilo = o.xl && o.zl;
  if (hilo) T = 4;
  a.tune_split = c->tune_split;
  // (hi+lo positions keep one granularity per launch: the mixed form was measured on plain fp32 positions only)
  a.tune_rsplit = (hilo && c->tune_sym_rsplit == 0) ? -2 : c->tune_sym_rsplit;
  a.shard_rank = (n_dev && sharded) ? c->shard_rank : 0;       // (host-sized launches get their tile block as arguments)
  a.shard_world = (n_dev && sharded) ? c->shard_world : 1;
  a.tail_items = c->sym_tail_items;
  const SymGeom gm = sym_geometry(n, T, a.tune_split, a.tune_rsplit, a.tail_items);
  a.ntiles = gm.ntiles;
  a.dmax = gm.dmax;
  a.i_first = i_first;
  a.i_count = i_count;
  a.ysplit = gm.ysplit;
  a.rsplit = gm.rsplit;
  a.ytail = gm.ytail;
  a.rbulk = gm.rbulk;
  a.xcd_run = c->xcd_run;
  a.acc_u = o.acc_u;
  a.acc_w = o.acc_w;
  a.scale = o.scale;
  a.bad = o.bad;
  a.vc4 = (float)vc4;
  // workgroups: 4 / rsplit items (tile, d-chunk) each
  long long blocks = sym_blocks(i_count, gm.ysplit, gm.rsplit, gm.ytail, gm.rbulk, c->xcd_run);
  if (n_dev) {
    const long long W = 64LL * T;
    for (long long nt = std::max<long long>(1, (std::max<long long>(n_lo, 1) + W - 1) / W); nt <= gm.ntiles; ++nt) {
      const SymGeom q = sym_geometry(nt * W, T, a.tune_split, gm.rsplit == 0 ? -1 : gm.rsplit, a.tail_items);    // the waves-per-item rule fixed by the bound: it picks the kernel
      blocks = std::max(blocks, sym_blocks(q.ntiles, q.ysplit, gm.rsplit, q.ytail, q.rbulk, c->xcd_run));
    }
  }
  if (!n_dev && i_count == 0) return LUDVM_OK;     // an owner without tiles (fewer tiles than owners)
  blocks = std::max<long long>(blocks, 1);         // (n_dev: the share is decided on the device; surplus waves leave)
  // Large launches: the quad variant (four I tiles of a workgroup share each partner tile: a quarter of the atomics) plus a
  // launch of the plain kernel restricted to the diagonal tiles.  The choice is a function of the vortex count (the march's
  // bound) alone, so every owner of a sharded ring makes the same one; owners must own whole quads.
  const bool one_wave_items = gm.rsplit == 1 || (gm.rsplit == 0 && gm.rbulk == 1);      // what the size rule gives at this size
  const bool quad = T == 8 && !hilo && gm.ntiles >= 16 &&
                    (c->tune_sym_rsplit == -4 || (c->sym_quad && one_wave_items && c->tune_sym_rsplit == 0 && gm.ntiles >= c->sym_quad_min_tiles));
  if (quad) {
    if (i_first % 4 != 0 || (i_count % 4 != 0 && i_first + i_count != gm.ntiles))
      return fail(c, LUDVM_E_ARG, "symmetric kernel, quad variant: an owner's tile block must start and end on multiples of 4 tiles");
    TimedLaunch tq{};
    bool act = false;
    CHK(timed_begin(c, tq, act));
    SymArgs d = a;
    d.diag_only = 1;
    const long long dblocks = std::max<long long>(1, sym_blocks(n_dev ? gm.ntiles : i_count, 1, 1, 0, 1, c->xcd_run));
    hipLaunchKernelGGL((pair_sym_f32<8, false, 1>), dim3((unsigned)dblocks), dim3(kBlock), 0, c->stream, d);
    const QuadGeom qg = quad_geometry<long long>(n, 8, a.tune_split);
    long long qblocks = quad_blocks(n_dev ? gm.ntiles : i_count, qg.ysplit, c->xcd_run);
    if (n_dev) {      // the device derives the chunks from its own vortex count: cover every count the bounds allow
      const long long W = 64LL * 8;
      for (long long nt = std::max<long long>(1, (std::max<long long>(n_lo, 1) + W - 1) / W); nt < gm.ntiles; ++nt)
        qblocks = std::max(qblocks, quad_blocks(nt, quad_geometry<long long>(nt * W, 8, a.tune_split).ysplit, c->xcd_run));
    }
    hipLaunchKernelGGL((pair_sym_quad_f32<8>), dim3((unsigned)std::max<long long>(qblocks, 1)), dim3(kBlock), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    CHK(timed_end(c, tq, act));
    return LUDVM_OK;
  }
  TimedLaunch t{};
  bool active = false;
  CHK(timed_begin(c, t, active));
  const dim3 grid((unsigned)blocks);
  const dim3 blk(kBlock);
#define LUDVM_SYM_LAUNCH(TT, HH)                                                                              \
  switch (gm.rsplit) {                                                                                        \
    case 0: hipLaunchKernelGGL((pair_sym_f32<TT, HH, 0>), grid, blk, 0, c->stream, a); break;                 \
    case 1: hipLaunchKernelGGL((pair_sym_f32<TT, HH, 1>), grid, blk, 0, c->stream, a); break;                 \
    case 2: hipLaunchKernelGGL((pair_sym_f32<TT, HH, 2>), grid, blk, 0, c->stream, a); break;                 \
    default: hipLaunchKernelGGL((pair_sym_f32<TT, HH, 4>), grid, blk, 0, c->stream, a); break;                \
  }
  if (hilo) { LUDVM_SYM_LAUNCH(4, true) }
  else if (T == 8) { LUDVM_SYM_LAUNCH(8, false) }
  else { LUDVM_SYM_LAUNCH(4, false) }
#undef LUDVM_SYM_LAUNCH
  HIPCHK(c, hipGetLastError());
  CHK(timed_end(c, t, active));
  return LUDVM_OK;
}

// The symmetric kernel's accumulators: [2 NaN counters | acc_u nt_pad | acc_w nt_pad] 64-bit integers, in the context's
// own buffer or in the caller's (ludvm_set_shard).  *acc points at acc_u; the counters sit at acc[-2], acc[-1].
int acc_buffer(ludvm_ctx* c, long long nt_pad, long long** acc) {
  const size_t bytes = ((size_t)2 * (size_t)nt_pad + 2) * sizeof(long long);
  if (c->ext_acc) {
    if (bytes > c->ext_acc_bytes) return fail(c, LUDVM_E_NOMEM, "the accumulator buffer given to ludvm_set_shard is too small");
    *acc = static_cast<long long*>(c->ext_acc) + 2;
    return LUDVM_OK;
  }
  CHK(ensure(c, c->acc, bytes));
  *acc = static_cast<long long*>(c->acc.p) + 2;
  return LUDVM_OK;
}

// Sum the accumulators (and their NaN counters) over all owners, in place and stream-ordered, before they are read.
bool sharded_at(const ludvm_ctx* c, long long n) { return (c->shard_world > 1 || c->comm_force) && n >= c->shard_min_n; }

// librccl, opened on first use (LUDVM_RCCL_LIB names the file; default librccl.so.1 from the loader's search path -- the
// copy a PyTorch-ROCm process already has mapped, if any).  Single-GPU users never load it.
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
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

// tile block of a shard owner (whole quads of 4 tiles: pair_sym_kernels.hpp, shard_block)
void shard_tiles(const ludvm_ctx* c, long long ntiles, long long* first, long long* count) {
  unsigned long long f, cnt;
  shard_block((unsigned long long)ntiles, c->shard_rank, c->shard_world, &f, &cnt);
  *first = (long long)f;
  *count = (long long)cnt;
}

SymScale* ctx_scale(ludvm_ctx* c) { return static_cast<SymScale*>(c->symsc.p); }
long long* ctx_bad(ludvm_ctx* c) { return reinterpret_cast<long long*>(static_cast<char*>(c->symsc.p) + 64); }

// Fixed-point scale record for circulations g[0, n) into (scale, bad); `partial` = workspace for the chunk sums.
int launch_sym_prepare(ludvm_ctx* c, const float* g, long long n, double vc4, SymScale* scale, long long* bad) {
  const long long nparts = (n + kPrepChunk - 1) / kPrepChunk;
  if (nparts <= 1) {
    hipLaunchKernelGGL(sym_prepare, dim3(1), dim3(kPrepBlock), 0, c->stream, g, n, vc4, scale, bad, (double*)nullptr);
  } else {
    CHK(ensure(c, c->symsc, 128 + (size_t)nparts * sizeof(double)));
    // (the record itself may live in c->symsc: re-derive the pointers after a grow)
    double* partial = reinterpret_cast<double*>(static_cast<char*>(c->symsc.p) + 128);
    hipLaunchKernelGGL(sym_prepare, dim3((unsigned)nparts), dim3(kPrepBlock), 0, c->stream, g, n, vc4, scale, bad, partial);
    hipLaunchKernelGGL(sym_prepare_final, dim3(1), dim3(64), 0, c->stream, partial, (int)nparts, vc4, scale, bad);
  }
  HIPCHK(c, hipGetLastError());
  return LUDVM_OK;
}

// Symmetric self-interaction of all of (x, z, g)[0, n) with the context's accumulators: zero them, derive the
// fixed-point scale from sum|Gamma| (unless the caller -- the march -- maintains it: scale / bad given), run the
// kernel.  The raw sums are left in c->acc as [acc_u | acc_w], each nt_pad 64-bit integers.
int launch_sym(ludvm_ctx* c, SymOperands o, long long n, double vc4, long long* nt_pad_out, const long long** acc_out,
               const long long** bad_out, const long long* n_dev = nullptr, long long n_lo = 0) {
  const long long nt_pad = (n + 63) / 64 * 64;
  long long* acc = nullptr;
  CHK(acc_buffer(c, nt_pad, &acc));
  CHK(ensure(c, c->symsc, 128 + (size_t)((n + kPrepChunk - 1) / kPrepChunk) * sizeof(double)));
  HIPCHK(c, hipMemsetAsync(acc - 2, 0, ((size_t)2 * (size_t)nt_pad + 2) * sizeof(long long), c->stream));
  o.acc_u = acc;
  o.acc_w = acc + nt_pad;
  if (!o.scale) {
    CHK(launch_sym_prepare(c, o.g, n, vc4, ctx_scale(c), ctx_bad(c)));
    o.scale = ctx_scale(c);
    o.bad = ctx_bad(c);
  }
  const bool sharded = sharded_at(c, n);      // (n: exact, or the march's bound -- the same number on every owner)
  if (sharded) o.bad = acc - 2;               // counted where the all-reduce sees it
  const int T = sym_tile_t(c, n, o.xl && o.zl, o.cx != nullptr);
  const long long ntiles = (n + 64LL * T - 1) / (64LL * T);
  long long first = 0, count = ntiles;
  if (sharded) shard_tiles(c, ntiles, &first, &count);
  CHK(launch_sym_tiles(c, T, o, n, first, count, vc4, n_dev, n_lo, sharded));
  if (sharded) CHK(reduce_accumulators(c, acc, nt_pad));
  *nt_pad_out = nt_pad;
  *acc_out = acc;
  *bad_out = o.bad;
  return LUDVM_OK;
}

// ---- spatial order of unordered inputs ----------------------------------------------------------------
// The fp32 kernels keep 1e-5 of max|u| because positions are offsets from the origin of a COMPACT origin class (256-element
// block x index parity).  A shed wake is compact in its stored order; a caller's array or a turbulence cloud
// (LUDVM.py:98-130) is not: 1.3e-4 / 5e-5 of max|u| for 1e5 / 1e6 vortices uniformly random in a 10 x 4 box at x = -55 with
// v_core = 1.3e-3 [MI355X, profiles/r04_unordered_accuracy.txt].  The reference's float64 sum (:565-569) does not depend on
// the order, so the host-pointer entry points may choose their own: Morton order, when -- and only when -- the given order
// is not already compact, so that a shed wake's bits are what they were.
constexpr size_t kOrderMin = 2048;     // below this many points on a side a stateless fp32 call does not run on local origins (see ludvm_induce_f64)
constexpr double kSmallSidePairsF64 = 268435456.0;   // 2^28 pairs: what float64 evaluates in ~0.2 ms [MI355X: 1.2-1.5e12 pairs/s]

struct OrderWs {
  unsigned* order[2];
  double* ext;
  double* sum;
  void* tmp;
  size_t tmp_bytes;
};

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

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
                     double* box = nullptr) {
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

// fp32 on local origins resolves a pair difference to ~6e-8 of its class's extent; next to a core of radius v_core that is
// up to ~5e-8 extent / v_core of max|u| for an area-filling cloud [MI355X, profiles/r04_extent_rule_calibration.txt: 2e5 ... 2e6
// vortices in a 10 x 4 box, v_core 6.5e-4 ... 6.5e-2, Morton order: errors 0.5 ... 5e-8 per unit of extent / v_core, a tail
// statistic of the rare pairs closer than v_core that straddle two classes; 9e-6 at a ratio of 189, 1.2e-5 at 376] and ~1.5e-8
// for a shed wake, whose close pairs follow each other in the stored order (3.5e-6 at config 2's 230).  Beyond these ratios
// -- a set too SPARSE for its core, which no order can mend: a class is 128 points wherever they lie -- a stateless fp32
// call takes hi+lo positions (exact differences, +30 % time) and a flow field float64, so LUDVM_PREC_F32 keeps 1e-5 of
// max|u| for any input.
constexpr double kMaxExtentOverCore = 300.0;          // the given order was kept (compact as stored: sheet-like)
constexpr double kMaxExtentOverCoreCloud = 150.0;     // the set had to be put in Morton order (area-filling)
inline bool too_sparse(double mean_extent, bool reordered, double vcore) {
  return vcore > 0.0 && mean_extent > (reordered ? kMaxExtentOverCoreCloud : kMaxExtentOverCore) * vcore;
}

bool valid_precision(int p) { return p == LUDVM_PREC_F32 || p == LUDVM_PREC_F32X2 || p == LUDVM_PREC_F64; }

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

}  // namespace

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
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);   // before its stream and buffers go
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
      hipLaunchKernelGGL(gather_f64, dim3(blocks_for((long long)ns)), dim3(kBlock), 0, c->stream, din, din + ns, din + 2 * ns, ord_s,
                         (long long)ns, din_ord, din_ord + ns, din_ord + 2 * ns);
      if (!self)
        hipLaunchKernelGGL(gather_f64, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream, din + 3 * ns, din + 3 * ns + nt,
                           (const double*)nullptr, ord_t, (long long)nt, din_ord + 3 * ns, din_ord + 3 * ns + nt, (double*)nullptr);
      HIPCHK(c, hipGetLastError());
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
      hipLaunchKernelGGL(scatter_f64, dim3(blocks_for((long long)nt)), dim3(kBlock), 0, c->stream, dout, dout + nt, ord_t, (long long)nt,
                         dout_ord, dout_ord + nt);
      HIPCHK(c, hipGetLastError());
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

// Roll-up launch on the resident wake (n vortices) with `nfoil` bound vortices already staged behind it
// at [n, n + nfoil) (masters and mirrors): pair kernel(s) + Euler finisher.  du/dw: optional device
// arrays receiving the induced velocities.  n_dev (march): the wake size is read on the device and `n` is
// only an upper bound that sizes the launch.
static inline unsigned fin_blocks(long long n) { return (unsigned)((n + kFinBlock - 1) / kFinBlock); }

// (march) where a symmetric launch sized from an upper bound finds its scale and how small the wake may be
struct MarchSym { const SymScale* scale = nullptr; long long* bad = nullptr; long long n_lo = 0; bool march = false; };

static int advect_launch(ludvm_ctx* c, size_t n, const long long* n_dev, double dt, size_t nfoil, double vcore,
                         int precision, double* du, double* dw, TailDuty td = TailDuty{}, MarchSym ms = MarchSym{}) {
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
      hipLaunchKernelGGL(gather_f64, gs_, bs, 0, c->stream, dxs, dzs, dgs, ord, (long long)ns, oxs, ozs, ogs);
      HIPCHK(c, hipGetLastError());
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

/* ---- measurement ---------------------------------------------------------------------------- */

#ifdef LUDVM_WAVE_TRACE
int ludvm_debug_set_wave_trace(ludvm_ctx* c, unsigned long long* d_trace) {
  if (!c) return LUDVM_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(ludvm::g_wave_trace), &d_trace, sizeof(d_trace)));
  return LUDVM_OK;
}
#endif

int ludvm_fixed_point_probe(ludvm_ctx* c, const float* values, size_t n, int scale_log2, long long* units) {
  if (!c) return LUDVM_E_ARG;
  if (n && (!values || !units)) return fail(c, LUDVM_E_ARG, "null array");
  if (scale_log2 < -120 || scale_log2 > 120) return fail(c, LUDVM_E_ARG, "scale_log2 must lie in [-120, 120]");
  if (n == 0) return LUDVM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  CHK(ensure(c, c->arena, Arena::need(n, 4) + Arena::need(n, 8)));
  Arena ar(c->arena.p);
  float* dv = ar.take<float>(n);
  long long* du = ar.take<long long>(n);
  HIPCHK(c, hipMemcpyAsync(dv, values, n * 4, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(fx_probe, dim3(blocks_for((long long)n)), dim3(kBlock), 0, c->stream, dv, (long long)n,
                     (float)std::ldexp(1.0, scale_log2), du);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(units, du, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LUDVM_OK;
}

int ludvm_kernel_timing(ludvm_ctx* c, int enable) {
  if (!c) return LUDVM_E_ARG;
  c->timing = enable != 0;
  return LUDVM_OK;
}

int ludvm_kernel_time_ms(ludvm_ctx* c, int reset, double* avg_ms, long long* launches) {
  if (!c) return LUDVM_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  CHK(drain_timing(c));
  if (avg_ms) *avg_ms = c->launches ? c->total_ms / (double)c->launches : 0.0;
  if (launches) *launches = c->launches;
  if (reset) {
    c->total_ms = 0.0;
    c->launches = 0;
  }
  return LUDVM_OK;
}

}  // extern "C"
