// Shared by the translation units of libludvm_hip.so: the context, error plumbing, the staging arena, and the internal
// host functions one unit offers the others.  Kernels are NOT here: templates and device helpers live in pair_kernels.hpp /
// pair_sym_kernels.hpp (includable anywhere); every non-template kernel lives in a header that exactly one unit includes
// (sym_prepare_kernels.hpp -> launch.hip, induce_kernels.hpp -> induce.hip, wake_kernels.hpp -> wake.hip, march_kernels.hpp ->
// march.hip, field_kernels.hpp -> flowfield.hip, order_kernels.hpp -> order.hip).
//
//   context.hip    lifecycle, streams, tuning, the error string, staging copies            (ludvm_create ... ludvm_set_symmetric)
//   launch.hip     plans and launches of the pair kernels, direct and symmetric; timing    (internal; ludvm_kernel_time_ms)
//   comm.hip       the library's own RCCL communicator                                     (ludvm_comm_*)
//   order.hip      spatial order of unordered inputs                                       (ludvm_spatial_order)
//   induce.hip     stateless pair sums, the multi-GPU shard step's entry points            (ludvm_induce_*, ludvm_advect_dev_f32, ludvm_sym_*)
//   wake.hip       the resident wake and its roll-up                                       (ludvm_wake_*)
//   march.hip      the device-resident time march                                          (ludvm_march_*)
//   flowfield.hip  flow-field grids and the vorticity stencil                              (ludvm_flowfield_*, ludvm_vorticity_*)
#pragma once
#pragma GCC visibility push(default)          // the C ABI is what the library exports; everything else is hidden (-fvisibility=hidden)
#include "../../include/ludvm_hip.h"
#pragma GCC visibility pop
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"
#include "march_types.hpp"
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

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};

struct TimedLaunch {
  hipEvent_t e0, e1;
};

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

namespace ludvm_host {

int fail(ludvm_ctx* c, int code, const std::string& msg);
int fail_hip(ludvm_ctx* c, const char* what, hipError_t e);

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

// ---- context.hip: device buffers and staging copies ---------------------------------------------------------------------
int ensure(ludvm_ctx* c, Buf& b, size_t bytes);
constexpr size_t kPinBytes = (size_t)1 << 20;
int h2d(ludvm_ctx* c, void* dst, const void* src, size_t bytes);
constexpr size_t kPinOutBytes = (size_t)1 << 16;
int d2h_small_sync(ludvm_ctx* c, const void* dsrc, size_t bytes, void** host_view);

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

inline bool valid_precision(int p) { return p == LUDVM_PREC_F32 || p == LUDVM_PREC_F32X2 || p == LUDVM_PREC_F64; }

// ---- launch.hip: plans and launches of the pair kernels -------------------------------------------------------------------
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
                                         // (64: more slabs than the shorter walks save, profiles/r06_march_chain_ab.txt)
constexpr long long kFewTargets = 256;
constexpr long long kTargetBlocks = 16384;  // total workgroups aimed for (2048 resident at 8/CU)
constexpr int kMaxSplit = 2048;

Plan make_plan(const ludvm_ctx* c, long long nt_launch, long long ns, int precision, bool small_ok = true, long long plan_nt = 0,
               bool grid_patch = false);
int timed_begin(ludvm_ctx* c, TimedLaunch& t, bool& active);
int timed_end(ludvm_ctx* c, TimedLaunch& t, bool active);
int drain_timing(ludvm_ctx* c);
int launch_pair(ludvm_ctx* c, PairArgs a, const Plan& p, int precision, void* u, void* w);
int induce_device(ludvm_ctx* c, const PairArgs& a, long long nt, long long ns, int precision, void* u, void* w,
                  long long plan_nt = 0);
long long sym_threshold(const ludvm_ctx* c, bool march = false);
bool use_symmetric(const ludvm_ctx* c, long long n, double vc4, bool march = false);
int sym_tile_t(const ludvm_ctx* c, long long n, bool hilo, bool local = false);

struct SymOperands {
  const float* x; const float* z; const float* g;
  const float* xl = nullptr; const float* zl = nullptr;     // hi+lo positions (T = 4)
  const float* cx = nullptr; const float* cz = nullptr;     // local origins: x, z are offsets from them
  long long* acc_u; long long* acc_w;
  const SymScale* scale; long long* bad;
};

int launch_sym_tiles(ludvm_ctx* c, int T, const SymOperands& o, long long n, long long i_first, long long i_count, double vc4,
                     const long long* n_dev = nullptr, long long n_lo = 0, bool sharded = false);
int acc_buffer(ludvm_ctx* c, long long nt_pad, long long** acc);
bool sharded_at(const ludvm_ctx* c, long long n);
void shard_tiles(const ludvm_ctx* c, long long ntiles, long long* first, long long* count);
inline SymScale* ctx_scale(ludvm_ctx* c) { return static_cast<SymScale*>(c->symsc.p); }
inline long long* ctx_bad(ludvm_ctx* c) { return reinterpret_cast<long long*>(static_cast<char*>(c->symsc.p) + 64); }
int launch_sym_prepare(ludvm_ctx* c, const float* g, long long n, double vc4, SymScale* scale, long long* bad);
int launch_sym(ludvm_ctx* c, SymOperands o, long long n, double vc4, long long* nt_pad_out, const long long** acc_out,
               const long long** bad_out, const long long* n_dev = nullptr, long long n_lo = 0);

// ---- comm.hip -----------------------------------------------------------------------------------------------------------
// Sum the accumulators (and their NaN counters) over all owners, in place and stream-ordered, before they are read.
int reduce_accumulators(ludvm_ctx* c, long long* acc, long long nt_pad);
// ludvm_destroy: the communicator goes before its stream and buffers do (no error to report to anyone)
void comm_release(ludvm_ctx* c);

// ---- order.hip ----------------------------------------------------------------------------------------------------------
constexpr size_t kOrderMin = 2048;     // below this many points on a side a stateless fp32 call does not run on local origins (see ludvm_induce_f64)
constexpr double kSmallSidePairsF64 = 268435456.0;   // 2^28 pairs: what float64 evaluates in ~0.2 ms [MI355X: 1.2-1.5e12 pairs/s]

struct OrderWs {
  unsigned* order[2];
  double* ext;
  double* sum;
  void* tmp;
  size_t tmp_bytes;
};

int order_workspace(ludvm_ctx* c, size_t nmax, OrderWs* w);
int class_extent_sum(ludvm_ctx* c, const OrderWs& w, const double* dx, const double* dz, const unsigned* order, size_t n, double* out,
                     double* box = nullptr);
int spatial_order_if_needed(ludvm_ctx* c, const OrderWs& w, int slot, const double* dx, const double* dz, size_t n,
                            const unsigned** order_out, double* mean_extent);
// d_k[i] = s_k[order[i]] for up to three float64 arrays (s2 / d2 may be NULL), and the way back for two: d_k[order[i]] = s_k[i]
int order_gather(ludvm_ctx* c, const double* s0, const double* s1, const double* s2, const unsigned* order, long long n, double* d0,
                 double* d1, double* d2);
int order_scatter(ludvm_ctx* c, const double* s0, const double* s1, const unsigned* order, long long n, double* d0, double* d1);
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

// ---- wake.hip -----------------------------------------------------------------------------------------------------------
int wake_grow(ludvm_ctx* c, size_t capacity);
int wake_refresh(ludvm_ctx* c, size_t first, size_t count);
inline unsigned fin_blocks(long long n) { return (unsigned)((n + kFinBlock - 1) / kFinBlock); }
// (march) where a symmetric launch sized from an upper bound finds its scale and how small the wake may be
struct MarchSym { const SymScale* scale = nullptr; long long* bad = nullptr; long long n_lo = 0; bool march = false; };
// Roll-up launch on the resident wake (n vortices) with `nfoil` bound vortices already staged behind it
// at [n, n + nfoil) (masters and mirrors): pair kernel(s) + Euler finisher.  du/dw: optional device
// arrays receiving the induced velocities.  n_dev (march): the wake size is read on the device and `n` is
// only an upper bound that sizes the launch.
int advect_launch(ludvm_ctx* c, size_t n, const long long* n_dev, double dt, size_t nfoil, double vcore,
                  int precision, double* du, double* dw, TailDuty td = TailDuty{}, MarchSym ms = MarchSym{});

}  // namespace ludvm_host

using namespace ludvm_host;
