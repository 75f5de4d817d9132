// libludvm_hip.so -- C-ABI implementation (see include/ludvm_hip.h for the contract and the reference file:line each entry point
// replaces; ctx.hpp for how the library is divided into translation units).  gfx950 only; no CPU path: every entry point either
// runs the HIP kernels or returns an error code.
// This unit: plans and launches of the pair kernels (direct with partial slabs, symmetric with fixed-point accumulators), the
// kernel stopwatch, the fixed-point probe.
#include "ctx.hpp"
#include "sym_prepare_kernels.hpp"

namespace ludvm_host {

// small_ok = false: the launch has no 256-source-tile kernel (generic flow-field grids), keep the chunk a multiple of 1024
// plan_nt: the target count that DECIDES the plan -- tile size, targets per lane and the split of the sources into
// partial sums, i.e. everything the rounding of a result depends on -- when the launch itself covers only a part of a
// larger target set (a block of rows of a flow-field grid: the block then carries the whole grid's bits); 0 = nt.
Plan make_plan(const ludvm_ctx* c, long long nt_launch, long long ns, int precision, bool small_ok, long long plan_nt, bool grid_patch) {
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
                  long long plan_nt) {
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
// Round 6 (profiles/r06_mid_size_variant_table.txt: every candidate forced at 22 sizes, ordered sheet, sustained load): between
// 35 000 and 44 000 vortices the two tiles alternate within +-2 % with the parity of their tile counts; the one size where the
// pick lost more (36 000: 512-vortex tiles 3.5 % behind) is what moved the switch from 34 816 to 36 864 = 72 tiles of 512.
constexpr long long kSymT8MinN = 36864;
static_assert(64 * 8 == LUDVM_SYM_TILE, "the multi-GPU entry points always use the 512-vortex tile");

// The symmetric kernel accumulates in fixed point, which needs the bound sum|Gamma| / (sqrt(2) v_core) on the raw
// sums: point vortices (v_core = 0, or so small that v_core^4 vanishes in fp32) take the direct kernel.
// In the march a symmetric step is an OVERLAPPED step: chord sums and solve run beside the kernel instead of in front
// of it (~40 us of a ~60 us serial step at 1e4 vortices), so it pays earlier there: from ~11 000 vortices [MI355X]
// (profiles/r02_march_symmetric_threshold.txt).
constexpr long long kSymMinNMarch = 11264;
long long sym_threshold(const ludvm_ctx* c, bool march) {
  return c->sym_mode == 1 ? (march ? kSymMinNMarch : kSymMinN) : (long long)c->sym_mode;
}
bool use_symmetric(const ludvm_ctx* c, long long n, double vc4, bool march) {
  if (c->sym_mode == 0 || !((float)vc4 > 0.0f)) return false;
  return n >= sym_threshold(c, march);
}

int sym_tile_t(const ludvm_ctx* c, long long n, bool hilo, bool local) {
  (void)local;                        // local origins fit both tiles (a 512-vortex tile keeps its targets twice)
  if (hilo) return 4;                 // hi+lo positions: 256-vortex tile only
  if (c->tune_sym_t == 4 || c->tune_sym_t == 8) return c->tune_sym_t;
  return n >= kSymT8MinN ? 8 : 4;
}

// Symmetric kernel over I tiles [i_first, i_first + i_count) of the tile ring of (x, z, g)[0, n); raw fixed-point
// sums are ADDED into acc_u / acc_w (n each, zeroed by the caller).  The partition of the work into partial sums
// is a function of n and T alone (sym_geometry).  n_dev (march): the vortex count is read on the device; it lies in
// [n_lo, n], and the grid is sized for the largest wave count any such n needs.
int launch_sym_tiles(ludvm_ctx* c, int T, const SymOperands& o, long long n, long long i_first, long long i_count, double vc4,
                     const long long* n_dev, long long n_lo, bool sharded) {
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

bool sharded_at(const ludvm_ctx* c, long long n) { return (c->shard_world > 1 || c->comm_force) && n >= c->shard_min_n; }

// tile block of a shard owner (whole quads of 4 tiles: pair_sym_kernels.hpp, shard_block)
void shard_tiles(const ludvm_ctx* c, long long ntiles, long long* first, long long* count) {
  unsigned long long f, cnt;
  shard_block((unsigned long long)ntiles, c->shard_rank, c->shard_world, &f, &cnt);
  *first = (long long)f;
  *count = (long long)cnt;
}

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
               const long long** bad_out, const long long* n_dev, long long n_lo) {
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

}  // namespace ludvm_host

extern "C" {

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
