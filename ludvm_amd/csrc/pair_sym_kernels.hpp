// Symmetric ("Newton's third law") Vatistas-core Biot-Savart kernel for SELF-interaction launches
// (targets == sources: wake roll-up LUDVM.py:1105-1124, BASELINE configs 3 and 4) on gfx950.
//
// The pair kernel is antisymmetric, K(i->j) = -K(j->i): dx, dz, r^2, q and -- the expensive part --
// rsq(q) are shared by the ordered pairs (i,j) and (j,i).  Direct evaluation costs 8 full-rate VALU
// ops + 1 v_rsq_f32 per ordered pair (24 issue cycles per 64 lanes on CDNA4); evaluating each
// UNORDERED pair once costs 11 ops + 1 rsq per TWO ordered pairs (15 cycles per ordered pair).
//
// Wave-autonomous systolic scheme (no LDS, no barriers, no cross-lane reductions):
//   * vortices are cut into tiles of W = 64*T; a wavefront owns one tile I: lane l keeps T targets
//     (x, z, Gamma, accumulators) in registers for its whole life;
//   * for each partner tile J the lane loads T vortices of J the same way, with their own
//     accumulators; lane l then evaluates its T x T pairs IN REGISTERS, updating both sides, and the
//     J set (x, z, Gamma and its accumulators) is rotated one lane with v_mov_b32_dpp wave_rol:1;
//     after 64 rotations every i of I has met every j of J and the J accumulators are back home;
//   * tile pairs are enumerated cyclically, J = I + d (mod NT), d = 1 .. (NT-1)/2, so every unordered
//     tile pair is visited once and every wave has the same amount of work; d = 0 (the diagonal tile,
//     which holds the self pairs) is evaluated with the ordered formula.  Because the work of a tile I
//     depends on I alone, a GPU that owns a contiguous block of I tiles does exactly 1/G of the job
//     (multi-GPU: the raw sums are then reduce-scattered, see ludvm_amd/sharded.py);
//   * results are accumulated into acc_u / acc_w (zeroed by the caller) as 64-BIT FIXED-POINT INTEGERS: each
//     wave's partial sum -- an fp32 number computed in a fixed order -- is scaled by a per-launch power of two,
//     truncated to an integer and added with global_atomic_add_x2.  Integer addition is associative, so the
//     result does not depend on the order in which the atomics land: a launch repeats BIT FOR BIT, and so do
//     the sums of several GPUs that split the tiles between them (the reference is deterministic too).  The
//     scale comes from the bound |sum| <= sum|Gamma| / (sqrt(2) v_core) of the Vatistas kernel (SymScale); a
//     finisher turns the integers into velocities (or an Euler step).  v_core = 0 has no such bound: inviscid
//     self-interaction launches use the direct kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "pair_kernels.hpp"

namespace ludvm {

// Fixed-point scale of a launch's raw sums (device memory; written by sym_prepare or by march_solve).
struct SymScale {
  float scale;   // a power of two: a partial sum p is accumulated as (long long)(p * scale)
  float pad;
  double inv;    // 1 / scale
};

struct SymArgs {
  const float* x; const float* z; const float* g;  // N vortices (device)
  const float* xl; const float* zl;                // lo parts of the positions (HILO kernels only)
  const float* cx; const float* cz;                // non-null: x, z are offsets from the origins cx, cz[i >> 8] of their
                                                   //   256-vortex blocks (local origins, pair_kernels.hpp)
  long long n;
  long long ntiles;      // ceil(n / (64*T))
  long long dmax;        // floor((ntiles-1)/2): symmetric offsets 1..dmax (+ ntiles/2 when even)
  long long i_first;     // this launch owns I tiles [i_first, i_first + i_count): one GPU's block of the
  long long i_count;     //   global tile ring (multi-GPU), or all tiles
  int ysplit;            // number of d-chunks
  int rsplit;            // 1, 2 or 4: the 64 rotation steps of a tile pair are shared by this many waves; 0: mixed --
  int ytail;             //   the items of the last `ytail` d-chunks by 4 waves, the others by `rbulk` (sym_geometry)
  int rbulk;
  int xcd_run;           // chunks per run of the XCD placement (0: ceil(chunks / 8); xcd_share / xcd_item)
  long long tail_items;  //   (how many single-wave items' worth of work that fine-grained end should hold)
  int diag_only;         // != 0: only the diagonal tiles (round -1 of the d-chunk-0 items): the quad variant's companion launch
  int tune_split, tune_rsplit;   // ludvm_set_tuning / ludvm_set_sym_tuning overrides (0 = heuristics), for n_dev launches
  long long* acc_u; long long* acc_w;   // raw fixed-point sums: u = acc_u / (scale 2 pi), w = -acc_w / (scale 2 pi)
  const SymScale* scale;
  long long* bad;        // incremented when a partial sum is not finite (NaN / inf inputs): the finisher then
                         //   returns NaN, as the reference's sum over all sources would
  float vc4;
  // Device-resident march (single GPU, all tiles): n is read from memory and ntiles / dmax / i_count / ysplit /
  // rsplit follow from it by the same rule the host uses (sym_geometry); the grid is sized from the host's upper
  // bound and surplus waves leave at once.
  const long long* n_dev;
  // Sharded roll-up of ONE simulation over several GPUs (ludvm_set_shard): with n_dev the owner's tile block
  // [i_first, i_first + i_count) is derived on the device from the tile count: tiles [NT r / G, NT (r + 1) / G)
  int shard_rank, shard_world;
};

// Launch geometry as a function of the vortex count alone (not of the owner's share, not of a host-side bound):
// the partition of the work into partial sums -- and with it every bit of the result -- is then the same for a
// march step sized from an upper bound, for one GPU and for G GPUs that own I-tile blocks of the same ring.
constexpr long long kSymTargetWaves = 8 * 65536;   // (I, d-chunk) work items aimed for over the whole ring
constexpr long long kSymMaxSplit = 64;
constexpr long long kSymMaxSplitTuned = 1024;      // what ludvm_set_tuning may ask for (measurements)
constexpr long long kXcds = 8;                        // XCDs of an MI355X: workgroup b is dispatched to XCD b % 8
constexpr long long kSymMaxRsplit = 4;
constexpr long long kSymMinItems = 10500;          // measured (profiles/r02_atomics_cost_and_lds_reduction.txt, table 4)
// Mixed granularity (rsplit = 0): a launch ends when its last waves do, and a launch of equal work items drains over about
// half an item's lifetime.  So the items that are dispatched LAST -- those of the highest d-chunks, in every owner's order
// -- are worked by four waves each (a quarter of the rotation steps per wave, partial sums added through LDS), the bulk
// before them by as many waves per item as the size rule gives (`rbulk` = 1 or 2; where the rule gives four there is
// nothing finer and the launch keeps one granularity).  Which items those are is a function of the vortex count alone (their
// d-chunk), so the partition into partial sums is the same for every owner of a sharded ring.  History [MI355X]: the first
// form (bulk always by single waves, 3072 items for the end, chunk counts that could leave the end empty) lost as often
// as it won (profiles/r03_mixed_granularity_negative_result.txt) and was shelved; with the bulk following the rule, no empty
// chunks and 1536 items for the end it is never slower than one granularity under sustained load and 1-5 % faster from
// 57 000 vortices up to the quad variant's range (profiles/r03_mixed_granularity_by_rule.txt), and is the default there.
// ludvm_set_sym_tuning(.., -1) / LUDVM_SYM_MIXED=1: at every size; -2 / LUDVM_SYM_MIXED=0: nowhere.
constexpr long long kSymTailItems = 1536;          // (half of 256 CUs x 4 SIMDs x 3 waves: measured, see above)
struct SymGeom { long long ntiles, dmax, dtot; int ysplit, rsplit, ytail, rbulk; };
// Placement of a launch's (unit, d-chunk) work items on the 8 XCDs (unit = I tile, or quad of I tiles): workgroup b runs
// on XCD b % 8.  The launch's `ys` d-chunks are cut into at most 8 RUNS of K = ceil(ys / 8) consecutive chunks, and in run r
// XCD x takes eighth (x + r) % 8 of the units.  Within a run an XCD works on neighbouring units whose ring offsets grow
// chunk by chunk, i.e. on overlapping partner tiles, which its L2 serves (round 2's point: memory-side fetches 2 GB ->
// 0.03 GB per N = 2^20 launch); and because the eighths ROTATE from run to run, every XCD meets every eighth once: all get
// the same number of items to within K - 1.  With a fixed eighth per XCD (rounds 2 and 3 until this) a unit count that is
// not a multiple of 8 left seven XCDs waiting for the eighth one: 129 tiles = 7 x 16 + 17, 6 % of the launch; 489 quads
// (N = 1e6) = 7 x 61 + 62, 1.4 % [MI355X].  (Rotating with EVERY chunk balances to within one item, and was measured
// first: same speed, but every chunk then meets new partner tiles: 1.0 GB of fetches per N = 1e6 launch instead of 0.03.)
struct XcdShare { unsigned lo, n; };
__host__ __device__ inline XcdShare xcd_share(unsigned units, unsigned e) {
  const unsigned lo = (unsigned)((unsigned long long)units * e / (unsigned)kXcds);
  return XcdShare{lo, (unsigned)((unsigned long long)units * (e + 1) / (unsigned)kXcds) - lo};
}
__host__ __device__ inline unsigned xcd_run(unsigned ys, int k) {
  return k > 0 ? (unsigned)k : (ys > 0 ? (ys + (unsigned)kXcds - 1) / (unsigned)kXcds : 1);
}
// items of XCD x in chunks [y0, y0 + ny) of a launch of ys chunks
__host__ __device__ inline unsigned long long xcd_items(unsigned units, unsigned ys, unsigned x, unsigned y0, unsigned ny,
                                                         int k = 0) {
  const unsigned K = xcd_run(ys, k);
  unsigned long long t = 0;
  for (unsigned y = y0; y < y0 + ny;) {
    const unsigned r = y / K, y_end = (r + 1) * K < y0 + ny ? (r + 1) * K : y0 + ny;
    t += (unsigned long long)xcd_share(units, (x + r) % (unsigned)kXcds).n * (y_end - y);
    y = y_end;
  }
  return t;
}
// item q of XCD x's list over chunks [y0, y0 + ny) (run by run, chunk-major within a run): its chunk and unit; false
// beyond the list
__host__ __device__ inline bool xcd_item(unsigned units, unsigned ys, unsigned x, unsigned y0, unsigned ny, unsigned q,
                                         unsigned& y_out, unsigned& unit, int k = 0) {
  const unsigned K = xcd_run(ys, k);
  y_out = y0; unit = 0;
  for (unsigned y = y0; y < y0 + ny;) {
    const unsigned r = y / K, y_end = (r + 1) * K < y0 + ny ? (r + 1) * K : y0 + ny;
    const XcdShare s = xcd_share(units, (x + r) % (unsigned)kXcds);
    const unsigned cnt = s.n * (y_end - y);
    if (q < cnt) {               // (s.n > 0 here)
      const unsigned c = q / s.n;
      y_out = y + c;
      unit = s.lo + (q - c * s.n);
      return true;
    }
    q -= cnt;
    y = y_end;
  }
  return false;
}
// Workgroups per XCD of a launch over i_count I tiles (the largest XCD's; surplus workgroups leave at once).  With
// rsplit = 0 an XCD's first workgroups hold 4 / rbulk bulk items each (d-chunks below ysplit - ytail), the rest one
// four-wave item each.
__host__ __device__ inline long long sym_blocks_xcd(long long i_count, long long ysplit, int rsplit, long long ytail,
                                                    int rbulk = 1, int k = 0) {
  long long most = 0;
  for (unsigned x = 0; x < (unsigned)kXcds; ++x) {
    long long wg;
    if (rsplit == 0) {
      const unsigned y1 = (unsigned)(ysplit - ytail);
      wg = ((long long)xcd_items((unsigned)i_count, (unsigned)ysplit, x, 0, y1, k) * rbulk + 3) / 4 +
           (long long)xcd_items((unsigned)i_count, (unsigned)ysplit, x, y1, (unsigned)ytail, k);
    } else {
      const long long ipb = 4 / rsplit;
      wg = ((long long)xcd_items((unsigned)i_count, (unsigned)ysplit, x, 0, (unsigned)ysplit, k) + ipb - 1) / ipb;
    }
    most = wg > most ? wg : most;
  }
  return most;
}
__host__ __device__ inline long long sym_blocks(long long i_count, long long ysplit, int rsplit, long long ytail = 0,
                                                int rbulk = 1, int k = 0) {
  return kXcds * sym_blocks_xcd(i_count, ysplit, rsplit, ytail, rbulk, k);
}

// Tile block of owner `rank` of `world` on a ring of ntiles tiles: whole quads of 4 consecutive tiles (the quad variant of
// the kernel adds the partial sums of a quad's four waves in fp32 before they are converted, so a quad must not be cut
// between two owners; the last block ends with the ring)
__host__ __device__ inline void shard_block(unsigned long long ntiles, int rank, int world, unsigned long long* first,
                                            unsigned long long* count) {
  const unsigned long long quads = (ntiles + 3) / 4;
  unsigned long long lo = 4 * (quads * (unsigned long long)rank / (unsigned long long)world);
  unsigned long long hi = 4 * (quads * (unsigned long long)(rank + 1) / (unsigned long long)world);
  if (lo > ntiles) lo = ntiles;
  if (hi > ntiles) hi = ntiles;
  *first = lo;
  *count = hi - lo;
}

// (I: long long on the host, unsigned on the device -- the kernel's prologue runs once per wave and a 64-bit division
// costs ~100 instructions there; both give the same numbers for n < 2^31)
template <typename I>
__host__ __device__ inline void sym_geometry_t(I n, int T, int tune_split, int tune_rsplit, I tail_items, I& ntiles, I& dmax,
                                               I& dtot, int& ysplit, int& rsplit, int& ytail, int& rbulk) {
  const I W = (I)(64 * T);
  ntiles = (n + W - 1) / W;
  dmax = ntiles > 0 ? (ntiles - 1) / 2 : 0;
  dtot = dmax + ((ntiles % 2 == 0 && ntiles > 1) ? 1 : 0);
  const I nt1 = ntiles > 0 ? ntiles : 1;
  // (kSymTargetWaves / nt1 rounded up is at least kSymMaxSplit whenever nt1 <= kSymTargetWaves / kSymMaxSplit: no division)
  I ys = tune_split > 0 ? (I)tune_split
                        : (nt1 <= (I)(kSymTargetWaves / kSymMaxSplit) ? (I)kSymMaxSplit : ((I)kSymTargetWaves + nt1 - 1) / nt1);
  if (ys > (I)kSymMaxSplit && tune_split <= 0) ys = (I)kSymMaxSplit;
  if (ys > (I)kSymMaxSplitTuned) ys = (I)kSymMaxSplitTuned;
  if (ys > dtot) ys = dtot;
  if (ys < 1) ys = 1;
  // no empty chunks: with `per` offsets per chunk, ceil(dtot / per) chunks cover the ring (96 offsets in 64 chunks would be 48
  // chunks of 2 and 16 empty ones -- waves that leave at once, and a mixed launch's fine-grained end without any work)
  if (dtot > 0) {
    const I per0 = (dtot + ys - 1) / ys;
    ys = (dtot + per0 - 1) / per0;
  }
  I rs = 1;
  ytail = 0;
  rbulk = 1;
  if (tune_rsplit == -4) {                     // the quad variant, whatever the size: single-wave geometry
    rs = 1;
  } else if (tune_rsplit == 1 || tune_rsplit == 2 || tune_rsplit == 4) {
    rs = (I)tune_rsplit;
  } else if (tune_rsplit == -1) {              // mixed: the last d-chunks by four waves per item
    rs = 0;
    const I yt = (tail_items + nt1 - 1) / nt1;
    ytail = (int)(yt > ys ? ys : yt);
    I rb = 1;                                  // the bulk before them by as many waves per item as the size rule gives
    while (rb < (I)kSymMaxRsplit && nt1 * ys * rb < (I)kSymMinItems) rb *= 2;
    rbulk = (int)rb;
  } else {                                     // by size: the smallest number of waves per item that gives enough work items
    while (rs < (I)kSymMaxRsplit && nt1 * ys * rs < (I)kSymMinItems) rs *= 2;
    // ... and, where that leaves room below four waves per item, the items dispatched last finer than the bulk (mixed
    // granularity).  tune_rsplit = -2: one granularity per launch, as until round 3
    if (tune_rsplit != -2 && rs < (I)kSymMaxRsplit) {
      rbulk = (int)rs;
      rs = 0;
      const I yt = (tail_items + nt1 - 1) / nt1;
      ytail = (int)(yt > ys ? ys : yt);
    }
  }
  ysplit = (int)ys;
  rsplit = (int)rs;
}
__host__ __device__ inline SymGeom sym_geometry(long long n, int T, int tune_split, int tune_rsplit,
                                                 long long tail_items = kSymTailItems) {
  SymGeom g;
  sym_geometry_t<long long>(n, T, tune_split, tune_rsplit, tail_items, g.ntiles, g.dmax, g.dtot, g.ysplit, g.rsplit, g.ytail,
                            g.rbulk);
  return g;
}

// acc += trunc(v * scale): order-independent accumulation of an fp32 partial sum (rounded toward zero to a whole unit of
// 1 / scale, i.e. 2^-61 of the bound on the sum: nine decimal digits below the fp32 rounding of the partial itself).
// The 64-bit integer is put together from two 32-bit conversions of the MAGNITUDE and negated as an integer for
// negative partials (11 VALU instructions; the compiler's float -> int64 sequence takes ~16, and there are 4 T of these
// per tile pair).  Every step is exact: t = v * scale (power-of-two scale), a = |t| < 2^63, h = floor(a / 2^32) < 2^31,
// and the remainder a - h 2^32 is an fp32 number in [0, 2^32) -- a multiple of ulp(a) below 2^32 has no more significant
// bits than a -- whose conversion to u32 drops at most a fraction.  (Round 2 split the SIGNED t: for t in (-2^31, 0) the
// remainder t + 2^32 is not an fp32 number and rounded by up to 128 units, up to 2^32 itself, whose conversion is
// undefined in C++.)
__device__ __forceinline__ unsigned long long fx_units(float v, float scale) {
  const float t = v * scale;
  const float a = __builtin_fabsf(t);
  const float h = __builtin_floorf(a * 2.3283064365386963e-10f);            // 2^-32
  const float r = __builtin_fmaf(-h, 4294967296.0f, a);
  const unsigned long long m = ((unsigned long long)(unsigned)h << 32) | (unsigned long long)(unsigned)r;
  const unsigned long long s = (unsigned long long)((long long)__builtin_bit_cast(int, t) >> 31);   // all ones for t < 0
  return (m ^ s) - s;                                                        // two's complement of -m when t < 0
}
__device__ __forceinline__ void fx_add(long long* acc, float v, float scale) {
  atomicAdd(reinterpret_cast<unsigned long long*>(acc), fx_units(v, scale));
}

// lane l receives the value of lane l+1 (wrapping): data moves one lane down
__device__ __forceinline__ float dpp_rol1(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x134, 0xf, 0xf, false));
}
__device__ __forceinline__ f32x2 dpp_rol1(f32x2 v) { return (f32x2){dpp_rol1(v.x), dpp_rol1(v.y)}; }
// The same move through the LDS crossbar (ds_bpermute_b32: no LDS memory involved) instead of the vector ALU: src4 = 4 *
// ((lane + 1) % 64).  A DPP move costs 4 VALU issue cycles on gfx950 and the rotation loop is VALU-bound (16 of them per
// rotation step of a 512-vortex tile = 3 % of its issue slots), so taking them off the VALU looked like 3 % -- measured
// [MI355X] it LOSES 3-4 % (N = 1e6: 122.8 ms against 118-120 ms with DPP on the same box, profiles/
// r03_rotation_through_lds_negative_result.txt): the accumulators come back after an LDS round trip that the next
// half-step's first J-side FMAs wait for.  Kept as a build switch (-DLUDVM_SYM_ROTATE_LDS=1), off.
#ifndef LUDVM_SYM_ROTATE_LDS
#define LUDVM_SYM_ROTATE_LDS 0
#endif
__device__ __forceinline__ float lds_rol1(float v, int src4) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f32x2 rol1(f32x2 v, int src4) {
  if (LUDVM_SYM_ROTATE_LDS) return (f32x2){lds_rol1(v.x, src4), lds_rol1(v.y, src4)};
  return dpp_rol1(v);
}

// Packed targets.  A v_pk_* instruction pairs two SOURCES against one target, so the target operand is the same number
// in both halves.  Instead of keeping every target twice ({x, x}: what hipcc does with a splat, re-made with v_mov
// where registers run short), a register pair holds TWO different targets and op_sel / op_sel_hi broadcast the wanted
// half: the same instruction, half the target registers.  The 512-vortex tile drops from 246 to 158 VGPRs (3 waves per
// SIMD instead of 2); the instruction count and the speed at the headline size are unchanged, 65-100 k vortices gain
// ~2 % (profiles/r02_packed_targets_ab.txt).  hipcc does not select these forms itself; tools/ubench/opsel_check.hip
// pins their semantics on the hardware.
// LUDVM_SYM_PACK=0 builds the splat form (same bits), LUDVM_SYM_OCC8 is the occupancy the T = 8 kernel is compiled for.
#ifndef LUDVM_SYM_PACK
#define LUDVM_SYM_PACK 1
#endif
#ifndef LUDVM_SYM_OCC8
#define LUDVM_SYM_OCC8 3
#endif
constexpr bool kPackTargets = LUDVM_SYM_PACK != 0;
// {p[half], p[half]} - s
__device__ __forceinline__ f32x2 pk_sub_sel(f32x2 p, f32x2 s, int half) {
  if (!kPackTargets) return (half ? (f32x2){p.y, p.y} : (f32x2){p.x, p.x}) - s;
  f32x2 d;
  if (half) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(s));
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(s));
  return d;
}
// s * {p[half], p[half]}.  `s` is the result of two v_rsq_f32: on gfx950 a VALU instruction may not read a
// transcendental result in the very next issue slot, and the compiler's hazard pass does not look inside inline
// assembly.  `after` must therefore be a value that ordinary code computed FROM s (the j-side strength s * gj): naming it
// as an operand puts at least that instruction between the v_rsq and this one.
__device__ __forceinline__ f32x2 pk_mul_sel(f32x2 s, f32x2 p, int half, f32x2 after) {
  if (!kPackTargets) return s * (half ? (f32x2){p.y, p.y} : (f32x2){p.x, p.x});
  f32x2 d;
  if (half) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(s), "v"(p), "v"(after));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(s), "v"(p), "v"(after));
  return d;
}

// p[i] for i < n, `pad` beyond: the load itself is unconditional (index clamped into the array), so the tile loads are
// straight-line code instead of one branch per element.  The array holds at least one element.
__device__ __forceinline__ float load_or(const float* p, unsigned i, unsigned n, float pad) {
  const float v = p[i < n ? i : 0u];
  return i < n ? v : pad;
}

// Slab helpers: a wave's LDS slab holds T floats per home lane and component, as T/4 planes of
// [96 slots][4 floats]: slots 0..63 are the home lanes, slots 64..95 repeat home lanes 0..31.  Every ds_read_b128 /
// ds_write_b128 of a wave covers consecutive 16-byte slots (conflict-free); `home4` is 4 * home lane.
// The repeated slots take the modulo out of the rotation loop: at rotation step k a lane reads the slot of home lane
// (lane + k) % 64, and for k = 32 s + kk (s = 0, 1; kk < 32) that is slot b + kk without wrap-around, b = lane for
// s = 0 and (lane + 32) % 64 for s = 1 -- a base register plus an offset that grows by 16 bytes per step, instead of an
// add, a mask and a shift per step (5 of the ~440 VALU instructions of a rotation step of the 512-vortex tile).
// (PLANE = 256: a plain [64][4] plane without the repeated slots, for data that is only read by its own lane)
constexpr int kPlane = 96 * 4;       // floats per plane
template <int T, int PLANE = kPlane>
__device__ __forceinline__ void slab_store(float* l, int home4, const float (&v)[T]) {
#pragma unroll
  for (int q = 0; q < T / 4; ++q) {
    const f32x4 V = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    *reinterpret_cast<f32x4*>(&l[q * PLANE + home4]) = V;
    if (PLANE > 256 && home4 < 128) *reinterpret_cast<f32x4*>(&l[q * PLANE + 256 + home4]) = V;
  }
}
template <int T, int PLANE = kPlane>
__device__ __forceinline__ void slab_load(const float* l, int home4, f32x2 (&out)[T / 2]) {
#pragma unroll
  for (int q = 0; q < T / 4; ++q) {
    const f32x4 V = *reinterpret_cast<const f32x4*>(&l[q * PLANE + home4]);
    out[2 * q] = (f32x2){V.x, V.y};
    out[2 * q + 1] = (f32x2){V.z, V.w};
  }
}

// T vortices per lane on both sides (tile = 64*T).  Per rotation step a lane evaluates T*T unordered
// pairs with 5.5*T*T packed ops + T*T v_rsq_f32; the J tile's (x, z, Gamma) sit in a wave-private LDS slab
// and are read with a per-lane rotating address (T/4 ds_read_b128 per component, off the VALU pipe); only
// the 2*T J-accumulator registers travel between lanes (v_mov_b32_dpp, 4 issue cycles each on gfx950).
// HILO: positions are hi+lo fp32 pairs, dx = (xh_i - xh_j) + (xl_i - xl_j), +4 packed ops
// per two unordered pairs; everything after the difference is plain fp32.
//
// Work items and waves.  An item is (tile I, d-chunk y); with rsplit = R > 1 (mid-size launches: too few tile pairs
// to keep every SIMD busy to the end) the 64 rotation steps of each of its tile pairs are shared by R waves OF ONE
// WORKGROUP (a workgroup holds 4 / R items).  Their partial sums -- R sets of 2 T values per lane for the J side of
// every tile pair, and for the I side at the end -- are added through LDS in a fixed order before they go to the
// accumulators: the atomics, whose 64-B requests at the memory side are what limits mid sizes [MI355X], are issued
// once per item instead of once per wave.  The partial sums travel in the waves' own slabs, which are dead between two
// tile pairs (a separate 32 KB buffer limited the 512-vortex tile to two workgroups per CU, and the compiler then
// scheduled its rotation loop into ~300 registers: profiles/r02_packed_targets_ab.txt), at the price of a second
// workgroup barrier per round.  Every wave of a workgroup runs the same number of tile-pair rounds (`per`), valid or
// not, so the barriers are uniform.
// RED = false (R = 1; hi+lo positions never use the 512-vortex tile): every wave adds its own partial sums.
#ifdef LUDVM_WAVE_TRACE
// measurement build only (tools/sym_wave_trace.py): every wave stores its start and end time (100 MHz clock)
__device__ unsigned long long* g_wave_trace = nullptr;
#endif
// R = 0: mixed granularity -- a workgroup holds either four single-wave items or one four-wave item (sym_geometry).
template <int T, bool HILO = false, int R = 1, bool RED = (R != 1)>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(T == 8 || HILO ? LUDVM_SYM_OCC8 : 4)))
pair_sym_f32(SymArgs a) {
  static_assert(T == 4 || T == 8, "T vortices per lane, read as T/4 ds_read_b128 per component");
  static_assert(R == 0 || R == 1 || R == 2 || R == 4, "waves per item");
  static_assert(RED == (R != 1), "one wave per item has nothing to reduce");
  constexpr int RR = R == 0 ? 4 : R;          // waves per item where several share one
  // vortex, tile and item numbers are 32-bit on the device (the launcher refuses n >= 2^31): per-lane addresses are a
  // scalar base plus a 32-bit offset register, and the prologue -- run once per wave, which at mid sizes lives for 16 or
  // 32 rotation steps only -- needs two 32-bit divisions instead of a dozen 64-bit ones
  unsigned n = (unsigned)a.n, ntiles = (unsigned)a.ntiles, dmax = (unsigned)a.dmax, i_first = (unsigned)a.i_first,
           i_count = (unsigned)a.i_count;
  int ysplit = a.diag_only ? 1 : a.ysplit, ytail = a.ytail, rbulk = a.rbulk;
  if (a.n_dev) {
    // (the instantiation -- tile and waves-per-item rule -- is what the host chose from its bound on n)
    n = (unsigned)*a.n_dev;
    unsigned dtot_;
    int rs_;
    sym_geometry_t<unsigned>(n, T, a.tune_split, R == 0 ? -1 : R, (unsigned)a.tail_items, ntiles, dmax, dtot_, ysplit, rs_, ytail,
                             rbulk);
    i_first = 0;
    i_count = ntiles;
    if (a.shard_world > 1) {
      unsigned long long f, cnt;
      shard_block(ntiles, a.shard_rank, a.shard_world, &f, &cnt);
      i_first = (unsigned)f;
      i_count = (unsigned)cnt;
    }
    if (a.diag_only) ysplit = 1;
  }
#ifdef LUDVM_WAVE_TRACE
  const unsigned long long trace_t0 = wall_clock64();
#endif
  constexpr int H = T / 2;
  constexpr int kWaves = kBlock / 64;
  constexpr int kComp = HILO ? 5 : 3;
  __shared__ __attribute__((aligned(16))) float slab[kWaves][kComp][(T / 4) * kPlane];
  // RED: between two tile pairs a wave's slab also carries its 2 T x 64 partial sums to the reducing waves (it is dead
  // then; 2 T <= kComp T), component c of wave w at slab[w][0][c * 64 + lane]
  static_assert(2 * T * 64 <= kComp * (T / 4) * kPlane, "partial sums fit the slab");
  // local origins: this wave's own T offsets per lane (x, z), written once: every pass over a partner tile re-refers the
  // targets from them (4 KB per wave at T = 8; the hi+lo kernels do not use local origins)
  __shared__ __attribute__((aligned(16))) float ioff[HILO ? 1 : kWaves][HILO ? 1 : 2][HILO ? 4 : 64 * T];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: tile indices stay scalar
  // XCD-aware placement (xcd_share / xcd_item above): XCD x works through its items chunk by chunk (the diagonal-tile items
  // first) in the order of its workgroups; waves that run on an XCD at the same time hold neighbouring I tiles with the same
  // offsets d, i.e. overlapping partner tiles J, and the L2 serves them: memory-side fetches of an N = 2^20 launch 2.04 GB ->
  // 0.03 GB at unchanged speed (the kernel is ALU-bound; profiles/r02_xcd_aware_mapping.txt).  The items, and with them the
  // partial sums, are the same whatever the placement.
  const unsigned xcd = blockIdx.x % (unsigned)kXcds, qb = blockIdx.x / (unsigned)kXcds;
  // item within the XCD, and whether this workgroup's waves share one item (wave-uniform, workgroup-uniform)
  unsigned yq = 0, unit = 0;
  bool active, shared;
  int rr = RR;                             // waves of this workgroup's items
  if constexpr (R == 0) {
    // bulk workgroups hold 4 / rbulk items of rbulk waves each (d-chunks below ysplit - ytail), the rest one four-wave item
    const unsigned y1 = (unsigned)(ysplit - ytail);
    const unsigned rb = (unsigned)rbulk, ipb = (unsigned)kWaves / rb;
    const unsigned n_bulk = (unsigned)xcd_items(i_count, (unsigned)ysplit, xcd, 0, y1, a.xcd_run);
    const unsigned nb1 = (n_bulk * rb + 3) / 4;                    // this XCD's bulk workgroups
    const bool tail = qb >= nb1;
    rr = tail ? 4 : (int)rb;
    shared = rr > 1;
    const unsigned q = tail ? qb - nb1 : qb * ipb + (unsigned)wv / rb;
    active = tail ? xcd_item(i_count, (unsigned)ysplit, xcd, y1, (unsigned)ytail, q, yq, unit, a.xcd_run)
                  : (q < n_bulk && xcd_item(i_count, (unsigned)ysplit, xcd, 0, y1, q, yq, unit, a.xcd_run));
    // a workgroup without any item leaves as a whole; so do idle single-wave items (they meet no barrier); the idle waves
    // of a workgroup that shares items stay for its barriers
    const bool wg_active = tail ? active : qb * ipb < n_bulk;
    if (!wg_active || (!shared && !active)) return;
  } else {
    shared = R > 1;
    const unsigned q = qb * (kWaves / R) + wv / R;
    active = xcd_item(i_count, (unsigned)ysplit, xcd, 0, (unsigned)ysplit, q, yq, unit, a.xcd_run);
  }
  const int r = shared ? wv % rr : 0;      // this wave's share of the rotation steps
  const int w0 = wv - r;                   // first wave of the item in the workgroup
  // single-wave items meet no barrier: idle waves leave
  if (R == 1 && !active) return;
  if (!active) { yq = 0; unit = 0; }
  const unsigned I = i_first + unit;
  const int y = (int)yq;
  // This wave does rotation steps [k_lo, k_hi) of every tile pair.  A J accumulator set that starts in lane l at step
  // k_lo belongs to home lane (l + k_lo) and, one lane per step, sits in lane (home - k_hi) after the last step.
  const int ksteps = shared ? 64 / rr : 64;
  const int k_lo = r * ksteps;
  const int k_hi = k_lo + ksteps;
  constexpr unsigned W = 64u * T;
  float* const lx = slab[wv][0];
  float* const lz = slab[wv][1];
  float* const lg = slab[wv][2];
  float* const lxl = slab[wv][HILO ? 3 : 0];
  float* const lzl = slab[wv][HILO ? 4 : 1];

  const bool even = (ntiles % 2 == 0) && ntiles > 1;
  const int dtot = (int)dmax + (even ? 1 : 0);
  const int per = a.diag_only ? 0 : (int)(((unsigned)dtot + (unsigned)ysplit - 1) / (unsigned)ysplit);
  const int d_lo = 1 + y * per;
  int d_hi = d_lo + per;  // exclusive
  if (d_hi > dtot + 1) d_hi = dtot + 1;

  // my targets.  The packed ops pair two SOURCES against one target; targets 2 h and 2 h + 1 share register pair h and
  // op_sel broadcasts the wanted half (kPackTargets).
  // Local origins: a vortex's offset is relative to the origin of its CLASS -- its 256-vortex block x its index parity
  // (pair_kernels.hpp).  Lane l holds vortices I W + l + 64 t: their index parity is the lane's, their block is t / 4.
  // A partner tile J is taken in NS = T / 4 passes, one per origin block q of J (plane q of the slab: 4 J vortices per
  // home lane), and at rotation step k a lane meets the J vortices of home lane l + k, whose index parity is the
  // lane's own when k is even and the other one when k is odd: the targets are therefore kept twice per pass,
  // xq / zq[0] referred to the origin of J's class (q, own parity) and xq / zq[1] to (q, other parity), and the rotation
  // loop alternates between them -- the same 11 packed ops + 2 rsq per two unordered pairs as plain fp32, and no more
  // target registers than the one-origin-per-block layout of round 2 needed for a 512-vortex tile.
  constexpr int NS = T / 4;
  f32x2 xp[H], zp[H], gp[H], au[T], aw[T], xpl[H], zpl[H];
  f32x2 xq[2][H], zq[2][H];
  {
    float x0[T], z0[T], g0[T], xl0[T], zl0[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const unsigned i = I * W + lane + 64u * t;
      const unsigned nl = active ? n : 0u;         // (an idle wave of a reducing workgroup holds padding only)
      x0[t] = load_or(a.x, i, nl, kPadPosF); z0[t] = load_or(a.z, i, nl, kPadPosF); g0[t] = load_or(a.g, i, nl, 0.0f);
      xl0[t] = HILO ? load_or(a.xl, i, nl, 0.0f) : 0.0f; zl0[t] = HILO ? load_or(a.zl, i, nl, 0.0f) : 0.0f;
      au[t] = (f32x2){0.f, 0.f}; aw[t] = (f32x2){0.f, 0.f};
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      xp[h] = (f32x2){x0[2 * h], x0[2 * h + 1]}; zp[h] = (f32x2){z0[2 * h], z0[2 * h + 1]};
      gp[h] = (f32x2){g0[2 * h], g0[2 * h + 1]};
      xpl[h] = (f32x2){xl0[2 * h], xl0[2 * h + 1]}; zpl[h] = (f32x2){zl0[2 * h], zl0[2 * h + 1]};
      xq[0][h] = xp[h]; xq[1][h] = xp[h]; zq[0][h] = zp[h]; zq[1][h] = zp[h];
    }
    if constexpr (!HILO) { slab_store<T, 256>(ioff[wv][0], lane * 4, x0); slab_store<T, 256>(ioff[wv][1], lane * 4, z0); }
  }
  const f32x2 vc4 = {a.vc4, a.vc4};
  const float fxs = a.scale->scale;
  // Local origins.  Origin records are wave-uniform data (the tile indices are scalars): they are fetched as scalars --
  // for the own tile once, for a partner tile together with its vortices -- and a lane picks the record of its index
  // parity when the targets are re-referred, so no global load sits between the passes.
  const bool local = !HILO && a.cx != nullptr;
  const bool pl = (lane & 1) != 0;
  // the two records (even / odd class) of origin block b, or zeros for a block past the last vortex (no record; its
  // slots are padding, any finite number serves)
  struct Org { float x0, x1, z0, z1; };      // (x, z) of the even class, of the odd class
  auto scalar = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  // (the loads: same address in every lane; the values become scalars -- `to_scalars` -- only where they are used, so that a
  // prefetch does not wait for its own loads)
  auto origin_records = [&](unsigned b) -> Org {
    const bool has = local && (b << kOriginShift) < n;
    const unsigned s0 = has ? 2 * b : 0;
    Org o{0.0f, 0.0f, 0.0f, 0.0f};
    if (has) { o.x0 = a.cx[s0]; o.x1 = a.cx[s0 + 1]; o.z0 = a.cz[s0]; o.z1 = a.cz[s0 + 1]; }
    return o;
  };
  auto to_scalars = [&](const Org& v) -> Org { return Org{scalar(v.x0), scalar(v.x1), scalar(v.z0), scalar(v.z1)}; };
  const unsigned blk_i = (I * W) >> kOriginShift;
  Org oi[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) oi[q] = to_scalars(origin_records(blk_i + q));
  // this lane's own origins (its index parity), per origin block of the tile
  float oix[NS], oiz[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) { oix[q] = pl ? oi[q].x1 : oi[q].x0; oiz[q] = pl ? oi[q].z1 : oi[q].z0; }
  // refer this lane's targets to the two origins (ojx / ojz: even, odd class) of one origin block of the partner tile:
  // xq / zq[0] to the class of the lane's own index parity, [1] to the other.  One rounding of (origin_I - origin_J) +
  // offset per target and class pair: neighbouring classes keep their relative precision, far ones do not need it.
  auto refer_targets = [&](const Org oj) {
    f32x2 xo[H], zo[H];
    slab_load<T, 256>(ioff[HILO ? 0 : wv][0], lane * 4, xo);
    slab_load<T, 256>(ioff[HILO ? 0 : wv][HILO ? 0 : 1], lane * 4, zo);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      // the partner class met at relative parity rr has index parity pl ^ rr
      const bool odd = pl != (rr != 0);
      const float jx = odd ? oj.x1 : oj.x0, jz = odd ? oj.z1 : oj.z0;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        // (targets 2 h and 2 h + 1 lie in the same origin block, h / 2)
        const float ddx = oix[h / 2] - jx, ddz = oiz[h / 2] - jz;
        xq[rr][h] = xo[h] + (f32x2){ddx, ddx};
        zq[rr][h] = zo[h] + (f32x2){ddz, ddz};
      }
    }
  };
  // the 4 J vortices (two packed pairs) that plane q of the slab holds for the home lane at byte position pos4 / 4
  auto load_plane = [&](const float* l, int q, int pos4, f32x2 (&out)[2]) {
    const f32x4 V = *reinterpret_cast<const f32x4*>(&l[q * kPlane + pos4]);
    out[0] = (f32x2){V.x, V.y};
    out[1] = (f32x2){V.z, V.w};
  };
  // NaN / inf cannot be represented in the integer accumulators: chk sums what this lane hands over for its OWN tile (the
  // I side); not finite <=> the launch met a non-finite input.  The I side alone sees every bad input: a vortex with a
  // non-finite position or strength is a source (J member) of its own tile's diagonal round, where it poisons the I-side
  // sums of that tile's lanes -- so the J-side partials need no check of their own.
  float chk = 0.0f;
  const int rot4 = ((lane + 1) & 63) * 4;   // lane l takes over from lane l + 1
  const int lane32x4 = ((lane + 32) & 63) * 4;

  // ---- tile pairs: each unordered pair once, both sides accumulated ------------------------------------------------
  // Round -1 (items of d-chunk 0 only) is the diagonal tile, J = I, which holds the self pairs: the same code, but over
  // the 64 rotation steps every ordered pair (i, j) of the tile is met from i's lane AND from j's, so the i side alone
  // is the complete ordered sum and the J-side accumulators are dropped.  (A separate ordered loop for it -- 8 instead
  // of 11 packed ops per two pairs on 1 of ~NT/2 tile pairs -- made the compiler carry ~60 more live registers through
  // the rotation loops.)  Rounds 0 .. per - 1 are the item's tile pairs; every wave of the workgroup runs all of them,
  // valid or not, because the reducing variants meet at two workgroup barriers per round (none in round -1).
  // The partner tile of a round is fetched one round ahead -- for the first round here, before anything waits on the own
  // tile -- so its latency lies under the previous round's rotation loops (or under the own tile's loads): a wave of a
  // mid-size launch lives for one tile pair's 16 or 32 rotation steps, and a microsecond of exposed latency per round is
  // several per cent of that.
  struct Round { unsigned J; bool diag, valid; };
  auto round_of = [&](int dd) -> Round {
    Round rd;
    rd.diag = dd < 0;
    const int d = rd.diag ? 0 : d_lo + dd;
    rd.valid = rd.diag || (active && dd < per && d < d_hi && !(even && d == dtot && I >= ntiles / 2));  // the half-way offset pairs each tile twice
    rd.J = I + (unsigned)d;
    if (rd.J >= ntiles) rd.J -= ntiles;
    return rd;
  };
  float pjx[T], pjz[T], pjg[T], pjxl[T], pjzl[T];
  Org poj[NS];
  auto fetch_tile = [&](const Round& rd) {
    // (an invalid round loads the own tile's addresses: harmless, unused)
    const unsigned Jt = rd.valid ? rd.J : I;
    if ((Jt + 1) * W <= n) {           // a whole tile (all but the last one): no guards (wave-uniform)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const unsigned j = Jt * W + lane + 64u * t;
        pjx[t] = a.x[j]; pjz[t] = a.z[j]; pjg[t] = a.g[j];
        pjxl[t] = HILO ? a.xl[j] : 0.0f; pjzl[t] = HILO ? a.zl[j] : 0.0f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const unsigned j = Jt * W + lane + 64u * t;
        pjx[t] = load_or(a.x, j, n, kPadPosF); pjz[t] = load_or(a.z, j, n, kPadPosF); pjg[t] = load_or(a.g, j, n, 0.0f);
        pjxl[t] = HILO ? load_or(a.xl, j, n, 0.0f) : 0.0f; pjzl[t] = HILO ? load_or(a.zl, j, n, 0.0f) : 0.0f;
      }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) poj[q] = origin_records(((Jt * W) >> kOriginShift) + q);
  };
  constexpr bool kPrefetch = R != 0;      // (the mixed-granularity variant has no registers to spare for it)
  const int dd0 = (active && y == 0) ? -1 : 0;
  if (kPrefetch) fetch_tile(round_of(dd0));
  for (int dd = dd0; dd < per; ++dd) {
    const Round rd = round_of(dd);
    if (!kPrefetch) fetch_tile(rd);
    const bool diag = rd.diag, valid = rd.valid;
    const unsigned J = rd.J;
    f32x2 bu[H], bw[H];
#pragma unroll
    for (int m = 0; m < H; ++m) { bu[m] = (f32x2){0.f, 0.f}; bw[m] = (f32x2){0.f, 0.f}; }
    Org oj[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) oj[q] = to_scalars(poj[q]);
    if (valid) {
      slab_store<T>(lx, lane * 4, pjx); slab_store<T>(lz, lane * 4, pjz); slab_store<T>(lg, lane * 4, pjg);
      if (HILO) { slab_store<T>(lxl, lane * 4, pjxl); slab_store<T>(lzl, lane * 4, pjzl); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the next round's partner tile: loads in flight during this round's rotation loops
    if (kPrefetch && dd + 1 < per) fetch_tile(round_of(dd + 1));
    if (valid) {

      // Pass q: my T targets against the 4 J vortices per home lane of J's origin block q (J accumulators bu / bw[2 q],
      // [2 q + 1]); every pass walks the wave's rotation steps [k_lo, k_hi), so all J accumulators end up with the same
      // home lane.  (Issuing the reads of step k+1 ahead of the arithmetic of step k measured no gain: with 3-5 waves
      // per SIMD the LDS latency is already covered.)
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        if (local) refer_targets(oj[q]);
        // steps k = 32 s + kk: slot (lane + 32 s) % 64 + kk of the plane, no wrap-around (slots 64..95 repeat 0..31)
        for (int k = k_lo; k < k_hi; k += 2) {
          const int pos0 = ((k & 32) ? lane32x4 : lane * 4) + (k & 31) * 4;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            // at step k this lane holds the accumulators of the J vortices whose home lane is (lane + k) % 64
            const int pos = pos0 + 4 * rr;
            f32x2 xj[2], zj[2], gj[2], xjl[2], zjl[2];
            load_plane(lx, q, pos, xj); load_plane(lz, q, pos, zj); load_plane(lg, q, pos, gj);
            if (HILO) { load_plane(lxl, q, pos, xjl); load_plane(lzl, q, pos, zjl); }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
              const int m = 2 * q + mm;
#pragma unroll
              for (int t = 0; t < T; ++t) {
                f32x2 dx = pk_sub_sel(HILO ? xp[t / 2] : xq[rr][t / 2], xj[mm], t & 1);
                f32x2 dz = pk_sub_sel(HILO ? zp[t / 2] : zq[rr][t / 2], zj[mm], t & 1);
                if (HILO) { dx = dx + pk_sub_sel(xpl[t / 2], xjl[mm], t & 1); dz = dz + pk_sub_sel(zpl[t / 2], zjl[mm], t & 1); }
                f32x2 r2 = dx * dx;
                r2 = __builtin_elementwise_fma(dz, dz, r2);
                const f32x2 qq = __builtin_elementwise_fma(r2, r2, vc4);
                const f32x2 sv = {__builtin_amdgcn_rsqf(qq.x), __builtin_amdgcn_rsqf(qq.y)};
                const f32x2 sj = sv * gj[mm];      // strength of j acting on i
                const f32x2 si = pk_mul_sel(sv, gp[t / 2], t & 1, sj);  // strength of i acting on j
                au[t] = __builtin_elementwise_fma(dz, sj, au[t]);
                aw[t] = __builtin_elementwise_fma(dx, sj, aw[t]);
                bu[m] = __builtin_elementwise_fma(dz, si, bu[m]);
                bw[m] = __builtin_elementwise_fma(dx, si, bw[m]);
              }
            }
            // hand the pass's J accumulators to the lane that meets the same J vortices next step (lane - 1)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) { bu[2 * q + mm] = rol1(bu[2 * q + mm], rot4); bw[2 * q + mm] = rol1(bw[2 * q + mm], rot4); }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();   // the slab is rewritten by the next tile
    }
    // after step k_hi - 1 and its rotation this lane holds the set of home lane (lane + k_hi) % 64 (with all 64
    // steps done: its own); j feels the opposite of what i feels.
    // home lane h holds vortices J*W + h + 64*t as packed elements t = 0..T-1; component c = 4 m + {0: u of element
    // 2m, 1: u of 2m+1, 2: w of 2m, 3: w of 2m+1}
    const int home = (lane + k_hi) & 63;
    if (diag) continue;        // (no barrier in this round)
#ifdef LUDVM_SYM_BARRIER_PROBE
    // measurement build only: what two workgroup barriers per round would cost the barrier-free single-wave items (the
    // price of sharing a partner tile's J-side sums among the four waves of a workgroup)
    if (!RED) { __syncthreads(); __syncthreads(); }
#endif
    if (!RED || !shared) {
      if (valid) {
#pragma unroll
        for (int m = 0; m < H; ++m) {
          const unsigned j0 = J * W + home + 64u * (2 * m), j1 = j0 + 64;
          if (j0 < n) { fx_add(&a.acc_u[j0], -bu[m].x, fxs); fx_add(&a.acc_w[j0], -bw[m].x, fxs); }
          if (j1 < n) { fx_add(&a.acc_u[j1], -bu[m].y, fxs); fx_add(&a.acc_w[j1], -bw[m].y, fxs); }
        }
      }
    } else {
      if (valid) {
#pragma unroll
        for (int m = 0; m < H; ++m) {
          lx[(4 * m + 0) * 64 + home] = bu[m].x; lx[(4 * m + 1) * 64 + home] = bu[m].y;
          lx[(4 * m + 2) * 64 + home] = bw[m].x; lx[(4 * m + 3) * 64 + home] = bw[m].y;
        }
      }
      __syncthreads();
      if (valid) {
        // the R waves of the item share the 2 T components; each adds the R partials in wave order
#pragma unroll 1
        for (int c = r; c < 2 * T; c += rr) {
          float v = 0.0f;
#pragma unroll
          for (int q = 0; q < RR; ++q)
            if (R != 0 || q < rr) v += (&slab[w0 + q][0][0])[c * 64 + lane];
          const unsigned j = J * W + lane + 64u * (2 * (c / 4) + (c & 1));
          if (j < n) fx_add((c & 2) ? &a.acc_w[j] : &a.acc_u[j], -v, fxs);
        }
      }
      __syncthreads();     // the slabs are rewritten by the next tile pair
    }
  }

  // ---- the I side ------------------------------------------------------------------------------------------------
  if (!RED || !shared) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const unsigned i = I * W + lane + 64u * t;
      if (i < n) {
        const float su = au[t].x + au[t].y, sw = aw[t].x + aw[t].y;
        fx_add(&a.acc_u[i], su, fxs); fx_add(&a.acc_w[i], sw, fxs);
        chk += su + sw;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) { lx[(2 * t) * 64 + lane] = au[t].x + au[t].y; lx[(2 * t + 1) * 64 + lane] = aw[t].x + aw[t].y; }
    __syncthreads();
    if (active) {
#pragma unroll 1
      for (int c = r; c < 2 * T; c += rr) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < RR; ++q)
          if (R != 0 || q < rr) v += (&slab[w0 + q][0][0])[c * 64 + lane];
        const unsigned i = I * W + lane + 64u * (c / 2);
        if (i < n) { fx_add((c & 1) ? &a.acc_w[i] : &a.acc_u[i], v, fxs); chk += v; }
      }
    }
  }
  if (!(__builtin_fabsf(chk) < __builtin_inff())) atomicAdd(reinterpret_cast<unsigned long long*>(a.bad), 1ULL);
#ifdef LUDVM_WAVE_TRACE
  if (g_wave_trace && (threadIdx.x & 63) == 0) {
    const long long wid = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    g_wave_trace[2 * wid] = trace_t0;
    g_wave_trace[2 * wid + 1] = wall_clock64();
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Quad variant (round 3; large launches): the four waves of a workgroup hold four CONSECUTIVE I tiles I0 .. I0 + 3 and
// meet the SAME partner tile J in every round (wave w at ring offset d_w = D - w, D = J - I0).  The partner tile is staged
// once per workgroup (each wave loads a quarter of it), and the four waves' J-side partial sums are added through LDS in
// wave order before ONE fixed-point atomic per J vortex and component leaves the workgroup: a quarter of the atomics of
// the single-wave kernel (whose every tile pair sends its own), i.e. of the launch's memory-side write traffic, for two
// workgroup barriers per round (measured cost of those, by themselves: 0.1-0.25 % at N = 1e6).  Wave w misses the offsets
// D - w outside 1 .. dtot at the two ends of the D range (three partly empty rounds per quad and launch), so the variant
// is used from ~1000 tiles up, where that is < 0.5 % of a quad's rounds.  The diagonal tiles (the self pairs) are left
// to a launch of pair_sym_f32 restricted to them (SymArgs::diag_only).  Owners of a sharded ring must own whole quads
// (blocks of 4 tiles): the fp32 sum of four partials is not the integer sum of their conversions.
// ---------------------------------------------------------------------------------------------------------------------
constexpr unsigned kQuad = 4;
constexpr unsigned kQuadSplit = 128;
struct QuadGeom { unsigned ntiles, dmax, dtot, Dtot; int ysplit, per, nlong, pshort; };
template <typename I>
__host__ __device__ inline QuadGeom quad_geometry(I n, int T, int tune_split) {
  QuadGeom g;
  const I W = (I)(64 * T);
  const I nt = (n + W - 1) / W;
  g.ntiles = (unsigned)nt;
  g.dmax = nt > 0 ? (unsigned)((nt - 1) / 2) : 0;
  g.dtot = g.dmax + ((nt % 2 == 0 && nt > 1) ? 1u : 0u);
  g.Dtot = g.dtot + (kQuad - 1);
  // d-chunks per quad.  An item of `per` rounds lives per x ~0.17 ms; a launch drains over about half the lifetime of the
  // items dispatched LAST, while every item pays ~4 % of one round for its own targets (loads, I-side conversions and
  // atomics).  Uniform chunks (measured at N = 1e6, same box: 64 chunks 112.0 ms, 128: 111.6, 256: 111.6;
  // tools/ab_quad_chunks.sh) cannot have both small; so the chunks TAPER: with u = Dtot / 128 rounded up, the first seven
  // eighths of the offsets go in chunks of 2 u rounds and the last eighth -- the highest chunk numbers, which every XCD
  // dispatches last -- in chunks of u / 4 (at least 1) rounds.  A function of the vortex count alone, like everything
  // about the partition.  ludvm_set_tuning(.., k > 0) asks for k uniform chunks instead (measurements).
  if (tune_split > 0) {
    unsigned ys = (unsigned)tune_split;
    if (ys > (unsigned)kSymMaxSplitTuned) ys = (unsigned)kSymMaxSplitTuned;
    if (ys > g.Dtot) ys = g.Dtot;
    if (ys < 1) ys = 1;
    g.per = (int)((g.Dtot + ys - 1) / ys);
    g.ysplit = (int)((g.Dtot + (unsigned)g.per - 1) / (unsigned)g.per);     // (no empty chunk at the end)
    g.nlong = g.ysplit;
    g.pshort = g.per;
    return g;
  }
  const unsigned u = (g.Dtot + kQuadSplit - 1) / kQuadSplit;
  const unsigned P = 2 * (u > 0 ? u : 1), ps = u / 4 > 0 ? u / 4 : 1;
  const unsigned nlong = (g.Dtot - g.Dtot / 8) / P;
  const unsigned rest = g.Dtot - nlong * P;
  g.per = (int)P;
  g.pshort = (int)ps;
  g.nlong = (int)nlong;
  g.ysplit = (int)(nlong + (rest + ps - 1) / ps);
  if (g.ysplit < 1) g.ysplit = 1;
  return g;
}
// ring offsets D in [lo, hi) of d-chunk yq of a quad
__host__ __device__ inline void quad_chunk(const QuadGeom& g, unsigned yq, int& lo, int& hi) {
  const bool lg = (int)yq < g.nlong;
  lo = 1 + (lg ? (int)yq * g.per : g.nlong * g.per + ((int)yq - g.nlong) * g.pshort);
  hi = lo + (lg ? g.per : g.pshort);
  if (hi > (int)g.Dtot + 1) hi = (int)g.Dtot + 1;
}
// workgroups of a quad launch over I tiles [i_first, i_first + i_count) (i_first a multiple of 4): one per (quad, d-chunk),
// the same number for each XCD
__host__ __device__ inline long long quad_blocks(long long i_count, int ysplit, int k = 0) {
  const unsigned quads = (unsigned)((i_count + kQuad - 1) / kQuad);
  long long most = 0;
  for (unsigned x = 0; x < (unsigned)kXcds; ++x) {
    const long long wg = (long long)xcd_items(quads, (unsigned)ysplit, x, 0, (unsigned)ysplit, k);
    most = wg > most ? wg : most;
  }
  return kXcds * most;
}

template <int T>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LUDVM_SYM_OCC8)))
pair_sym_quad_f32(SymArgs a) {
  static_assert(T == 8, "the quad variant is built for the 512-vortex tile");
  constexpr int H = T / 2, NS = T / 4;
  constexpr unsigned W = 64u * T;
  constexpr int kWaves = kBlock / 64;
  static_assert(kWaves == (int)kQuad, "one I tile per wave");
  __shared__ __attribute__((aligned(16))) float slab[3][(T / 4) * kPlane];     // the partner tile: x, z, Gamma
  __shared__ __attribute__((aligned(16))) float ex[kWaves][2 * T * 64];        // the waves' J-side partial sums
  __shared__ __attribute__((aligned(16))) float ioff[kWaves][2][64 * T];       // own offsets (local origins)

  unsigned n = (unsigned)a.n, i_first = (unsigned)a.i_first, i_count = (unsigned)a.i_count;
  if (a.n_dev) {
    n = (unsigned)*a.n_dev;
    const unsigned nt = (n + W - 1) / W;
    i_first = 0;
    i_count = nt;
    if (a.shard_world > 1) {       // owners own whole quads
      unsigned long long f, cnt;
      shard_block(nt, a.shard_rank, a.shard_world, &f, &cnt);
      i_first = (unsigned)f;
      i_count = (unsigned)cnt;
    }
  }
  const QuadGeom gm = quad_geometry<unsigned>(n, T, a.tune_split);
  const unsigned ntiles = gm.ntiles, dmax = gm.dmax;
  const int dtot = (int)gm.dtot;
  const bool even = (ntiles % 2 == 0) && ntiles > 1;

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // XCD-aware placement over quads (as pair_sym_f32 over tiles): one workgroup per (quad, d-chunk)
  const unsigned xcd = blockIdx.x % (unsigned)kXcds, qb = blockIdx.x / (unsigned)kXcds;
  const unsigned quads = (i_count + kQuad - 1) / kQuad;
  unsigned yq, quad;
  if (!xcd_item(quads, (unsigned)gm.ysplit, xcd, 0, (unsigned)gm.ysplit, qb, yq, quad, a.xcd_run)) return;     // (the whole workgroup)
  const unsigned I0 = i_first + kQuad * quad;
  const unsigned I = I0 + (unsigned)wv;
  const bool mine = I < i_first + i_count && I < ntiles;       // this wave's tile exists and is this owner's
  int D_lo, D_hi;
  quad_chunk(gm, yq, D_lo, D_hi);
  const int per = D_hi - D_lo;                                 // this workgroup's rounds

  float* const lx = slab[0];
  float* const lz = slab[1];
  float* const lg = slab[2];

  // ---- my targets (as pair_sym_f32) ----------------------------------------------------------------------------------
  f32x2 gp[H], au[T], aw[T];
  f32x2 xq[2][H], zq[2][H];
  {
    float x0[T], z0[T], g0[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const unsigned i = I * W + lane + 64u * t;
      const unsigned nl = mine ? n : 0u;
      x0[t] = load_or(a.x, i, nl, kPadPosF); z0[t] = load_or(a.z, i, nl, kPadPosF); g0[t] = load_or(a.g, i, nl, 0.0f);
      au[t] = (f32x2){0.f, 0.f}; aw[t] = (f32x2){0.f, 0.f};
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      gp[h] = (f32x2){g0[2 * h], g0[2 * h + 1]};
      xq[0][h] = (f32x2){x0[2 * h], x0[2 * h + 1]}; xq[1][h] = xq[0][h];
      zq[0][h] = (f32x2){z0[2 * h], z0[2 * h + 1]}; zq[1][h] = zq[0][h];
    }
    slab_store<T, 256>(ioff[wv][0], lane * 4, x0); slab_store<T, 256>(ioff[wv][1], lane * 4, z0);
  }
  const f32x2 vc4 = {a.vc4, a.vc4};
  const float fxs = a.scale->scale;
  const bool local = a.cx != nullptr;
  const bool pl = (lane & 1) != 0;
  struct Org { float x0, x1, z0, z1; };
  auto scalar = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  auto origin_records = [&](unsigned b) -> Org {
    const bool has = local && (b << kOriginShift) < n;
    const unsigned s0 = has ? 2 * b : 0;
    Org o{0.0f, 0.0f, 0.0f, 0.0f};
    if (has) { o.x0 = a.cx[s0]; o.x1 = a.cx[s0 + 1]; o.z0 = a.cz[s0]; o.z1 = a.cz[s0 + 1]; }
    return o;
  };
  auto to_scalars = [&](const Org& v) -> Org { return Org{scalar(v.x0), scalar(v.x1), scalar(v.z0), scalar(v.z1)}; };
  float oix[NS], oiz[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const Org o = to_scalars(origin_records(mine ? ((I * W) >> kOriginShift) + q : 0xffffffffu >> kOriginShift));
    oix[q] = pl ? o.x1 : o.x0; oiz[q] = pl ? o.z1 : o.z0;
  }
  auto refer_targets = [&](const Org oj) {
    f32x2 xo[H], zo[H];
    slab_load<T, 256>(ioff[wv][0], lane * 4, xo);
    slab_load<T, 256>(ioff[wv][1], lane * 4, zo);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const bool odd = pl != (rr != 0);
      const float jx = odd ? oj.x1 : oj.x0, jz = odd ? oj.z1 : oj.z0;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float ddx = oix[h / 2] - jx, ddz = oiz[h / 2] - jz;
        xq[rr][h] = xo[h] + (f32x2){ddx, ddx};
        zq[rr][h] = zo[h] + (f32x2){ddz, ddz};
      }
    }
  };
  auto load_plane = [&](const float* l, int q, int pos4, f32x2 (&out)[2]) {
    const f32x4 V = *reinterpret_cast<const f32x4*>(&l[q * kPlane + pos4]);
    out[0] = (f32x2){V.x, V.y};
    out[1] = (f32x2){V.z, V.w};
  };
  const int lane32x4 = ((lane + 32) & 63) * 4;
  float chk = 0.0f;

  // ---- this wave's quarter of a partner tile, fetched one round ahead: the four slices of slab plane wv / 2 for the home
  //      lanes of half wv & 1 -- lane l holds slices 4 (wv / 2) + 2 (l & 1), + 1 of home lane 32 (wv & 1) + l / 2 ------------
  const int qhome = 32 * (wv & 1) + (lane >> 1);
  float pjx[2], pjz[2], pjg[2];
  Org poj[NS];
  auto partner_of = [&](int dd) -> unsigned {
    unsigned J = I0 + (unsigned)(D_lo + dd);
    if (J >= ntiles) J -= ntiles;
    return J;
  };
  auto fetch_quarter = [&](int dd) {
    const unsigned J = (D_lo + dd < D_hi) ? partner_of(dd) : I0;      // (beyond the chunk: any valid tile, unused)
    const unsigned Jc = J < ntiles ? J : 0u;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned j = Jc * W + (unsigned)qhome + 64u * (unsigned)(4 * (wv >> 1) + 2 * (lane & 1) + s);
      pjx[s] = load_or(a.x, j, n, kPadPosF); pjz[s] = load_or(a.z, j, n, kPadPosF); pjg[s] = load_or(a.g, j, n, 0.0f);
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) poj[q] = origin_records(((Jc * W) >> kOriginShift) + q);
  };
  // A lane's pair is components 2 (lane & 1), + 1 of slot `qhome` in plane wv / 2: byte 8 * lane of the wave's 512-byte run
  // -- consecutive 8-byte stores, no bank conflict.  (Round 3 gave wave w slices 2 w, 2 w + 1 of EVERY home lane: 8-byte
  // stores 16 bytes apart, two to four lanes per bank -- SQ_LDS_BANK_CONFLICT 3.4e7 cycles per N = 1e6 launch.)  The slab's
  // contents, and with them every result bit, are the same.
  auto store_quarter = [&]() {
    const int at = (wv >> 1) * kPlane + 128 * (wv & 1) + 2 * lane;
    auto put = [&](float* l, const float (&v)[2]) {
      *reinterpret_cast<f32x2*>(&l[at]) = (f32x2){v[0], v[1]};
      if ((wv & 1) == 0) *reinterpret_cast<f32x2*>(&l[at + 256]) = (f32x2){v[0], v[1]};      // slots 64 .. 95 repeat 0 .. 31
    };
    put(lx, pjx); put(lz, pjz); put(lg, pjg);
  };

  fetch_quarter(0);
  for (int dd = 0; dd < per; ++dd) {
    const int D = D_lo + dd;
    if (D >= D_hi) break;                                       // (uniform over the workgroup)
    const unsigned J = partner_of(dd);
    // wave w meets J at ring offset D - w: one of its offsets 1 .. dmax, or the half-way offset of an even ring (lower half)
    auto valid_of = [&](int w) -> bool {
      const int d = D - w;
      const unsigned Iw = I0 + (unsigned)w;
      const bool own = Iw < i_first + i_count && Iw < ntiles;
      return own && ((d >= 1 && d <= (int)dmax) || (even && d == dtot && Iw < ntiles / 2));
    };
    const bool valid = valid_of(wv);
    const bool any = valid_of(0) || valid_of(1) || valid_of(2) || valid_of(3);
    if (any) store_quarter();
    Org oj[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) oj[q] = to_scalars(poj[q]);
    __syncthreads();                                            // the partner tile is complete (and the previous round's ex read)
    if (dd + 1 < per) fetch_quarter(dd + 1);
    if (!any) continue;                                         // (uniform: nobody reads the slab, nothing to hand over)
    f32x2 bu[H], bw[H];
#pragma unroll
    for (int m = 0; m < H; ++m) { bu[m] = (f32x2){0.f, 0.f}; bw[m] = (f32x2){0.f, 0.f}; }
    if (valid) {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        if (local) refer_targets(oj[q]);
        for (int k = 0; k < 64; k += 2) {
          const int pos0 = ((k & 32) ? lane32x4 : lane * 4) + (k & 31) * 4;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int pos = pos0 + 4 * rr;
            f32x2 xj[2], zj[2], gj[2];
            load_plane(lx, q, pos, xj); load_plane(lz, q, pos, zj); load_plane(lg, q, pos, gj);
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
              const int m = 2 * q + mm;
#pragma unroll
              for (int t = 0; t < T; ++t) {
                const f32x2 dx = pk_sub_sel(xq[rr][t / 2], xj[mm], t & 1);
                const f32x2 dz = pk_sub_sel(zq[rr][t / 2], zj[mm], t & 1);
                f32x2 r2 = dx * dx;
                r2 = __builtin_elementwise_fma(dz, dz, r2);
                const f32x2 qq = __builtin_elementwise_fma(r2, r2, vc4);
                const f32x2 sv = {__builtin_amdgcn_rsqf(qq.x), __builtin_amdgcn_rsqf(qq.y)};
                const f32x2 sj = sv * gj[mm];
                const f32x2 si = pk_mul_sel(sv, gp[t / 2], t & 1, sj);
                au[t] = __builtin_elementwise_fma(dz, sj, au[t]);
                aw[t] = __builtin_elementwise_fma(dx, sj, aw[t]);
                bu[m] = __builtin_elementwise_fma(dz, si, bu[m]);
                bw[m] = __builtin_elementwise_fma(dx, si, bw[m]);
              }
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) { bu[2 * q + mm] = dpp_rol1(bu[2 * q + mm]); bw[2 * q + mm] = dpp_rol1(bw[2 * q + mm]); }
          }
        }
      }
    }
    // after the 64 steps every lane holds the sums of its own home lane: J vortex J W + lane + 64 t, packed elements
    // t = 0 .. T - 1; component c = 4 m + {0: u of 2 m, 1: u of 2 m + 1, 2: w of 2 m, 3: w of 2 m + 1}
#pragma unroll
    for (int m = 0; m < H; ++m) {
      ex[wv][(4 * m + 0) * 64 + lane] = bu[m].x; ex[wv][(4 * m + 1) * 64 + lane] = bu[m].y;
      ex[wv][(4 * m + 2) * 64 + lane] = bw[m].x; ex[wv][(4 * m + 3) * 64 + lane] = bw[m].y;
    }
    __syncthreads();                                            // every wave has left the slab and handed its sums over
    // the four waves share the 2 T components; each adds the four partials in wave order (zeros from waves without a pair)
#pragma unroll 1
    for (int c = wv; c < 2 * T; c += kWaves) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) v += ex[w][c * 64 + lane];
      const unsigned j = J * W + lane + 64u * (2 * (c / 4) + (c & 1));
      if (j < n) fx_add((c & 2) ? &a.acc_w[j] : &a.acc_u[j], -v, fxs);      // j feels the opposite of what i feels
    }
  }

  // ---- the I side ------------------------------------------------------------------------------------------------
  if (mine) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const unsigned i = I * W + lane + 64u * t;
      if (i < n) {
        const float su = au[t].x + au[t].y, sw = aw[t].x + aw[t].y;
        fx_add(&a.acc_u[i], su, fxs); fx_add(&a.acc_w[i], sw, fxs);
        chk += su + sw;
      }
    }
  }
  if (!(__builtin_fabsf(chk) < __builtin_inff())) atomicAdd(reinterpret_cast<unsigned long long*>(a.bad), 1ULL);
}

// One workgroup: the fixed-point scale of a launch from sum |Gamma| (summed in a fixed order -> the same bits every
// time, on every GPU that holds the same array).  |raw sum| <= sum|Gamma| max_r r / sqrt(r^4 + vc^4) = sum|Gamma| /
// (sqrt(2) vc); the scale leaves a factor 4 of headroom under 2^63 on top of that.  Also clears the launch's NaN flag.
constexpr int kPrepBlock = 1024;
__device__ __forceinline__ void sym_scale_from_sum(double sum_abs, double vc4, SymScale* out, long long* bad) {
  const double vc = sqrt(sqrt(vc4));
  const double bound = sum_abs / (1.4142135623730951 * vc);
  int e = 0;
  if (bound > 0.0 && bound < 1.0e300) (void)frexp(bound, &e);    // bound < 2^e
  else if (!(bound == 0.0) && bad) atomicAdd(reinterpret_cast<unsigned long long*>(bad), 1ULL);   // NaN / inf strengths
  int k = 61 - e;
  if (k > 120) k = 120;
  if (k < -120) k = -120;
  out->scale = (float)ldexp(1.0, k);
  out->pad = 0.0f;
  out->inv = ldexp(1.0, -k);
}

// sum |g| over [first, first + count) by one workgroup, in a fixed order: 8 independent running sums per thread
// (8 loads in flight), a fixed shuffle tree per wavefront, the wave partials in order.  Every thread returns it.
__device__ __forceinline__ double block_abs_sum(const float* g, long long first, long long count) {
  __shared__ double part[kPrepBlock / 64];
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long end = first + count;
  long long i = first + threadIdx.x;
  for (; i + 7 * kPrepBlock < end; i += 8 * kPrepBlock) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += (double)__builtin_fabsf(g[i + k * kPrepBlock]);
  }
  for (int k = 0; i < end; i += kPrepBlock, ++k) s[k & 7] += (double)__builtin_fabsf(g[i]);
  double t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
  __syncthreads();
  double tot = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += part[w];
  return tot;
}

// Fixed-point scale of a launch from sum |Gamma|.  Up to kPrepChunk vortices: one workgroup does it all.  Beyond:
// gridDim.x workgroups each sum a contiguous chunk into partial[blockIdx.x] and sym_prepare_final adds those in order.
constexpr long long kPrepChunk = 65536;

// fixed-point raw sum -> fp32 raw sum (NaN when the launch met a non-finite partial sum)
__device__ __forceinline__ float fx_read(const long long* acc, long long i, const SymScale* sc, bool bad) {
  if (bad) return __builtin_nanf("");
  return (float)((double)acc[i] * sc->inv);
}
__device__ __forceinline__ float fx_read(const long long* acc, long long i, const SymScale* sc, const long long* bad) {
  return fx_read(acc, i, sc, *bad != 0);
}

// What `nsrc` extra sources staged in the float64 master arrays at [first, first + nsrc) -- the bound vortices of a
// time step (LUDVM.py:1106, :1115, :1124), in the overlapped march also the vortices just shed -- induce on this
// thread's vortex at (xi, zi): a few hundred sources, so the sum is done right in the Euler finisher, one target
// per thread.  Positions are taken from the masters relative to the first staged source (the airfoil: what is
// near it keeps full precision), circulations from g32.  Every thread of the workgroup must call this.
constexpr int kFoilChunk = 256;
__device__ __forceinline__ void staged_sources_on(bool on, double xi, double zi, const double* x64, const double* z64,
                                                  const float* g32, long long first, int nsrc, float vc4, float& fu,
                                                  float& fw, int nextra = 0, const double* extra = nullptr) {
  // `extra` = [x0, x1, z0, z1, g0, g1]: up to two more sources kept outside the wake arrays (the vortices shed in
  // an overlapped march step: their wake entries are being moved by the very kernel that calls this)
  __shared__ float fx[kFoilChunk], fz[kFoilChunk], fg[kFoilChunk];
  fu = 0.0f; fw = 0.0f;
  const int ntot = nsrc + nextra;
  if (ntot <= 0) return;                       // uniform over the workgroup
  const double ax = nsrc > 0 ? x64[first] : extra[0], az = nsrc > 0 ? z64[first] : extra[2];
  const float xr = (float)(xi - ax), zr = (float)(zi - az);
  for (int base = 0; base < ntot; base += kFoilChunk) {
    const int cnt = ntot - base < kFoilChunk ? ntot - base : kFoilChunk;
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
      const int q = base + t;
      if (q < nsrc) {
        fx[t] = (float)(x64[first + q] - ax);
        fz[t] = (float)(z64[first + q] - az);
        fg[t] = g32[first + q];
      } else {
        fx[t] = (float)(extra[q - nsrc] - ax);
        fz[t] = (float)(extra[2 + q - nsrc] - az);
        fg[t] = (float)extra[4 + q - nsrc];
      }
    }
    __syncthreads();
    if (on) {
      for (int j = 0; j < cnt; ++j) {
        const float dx = xr - fx[j], dz = zr - fz[j];
        const float r2 = __builtin_fmaf(dz, dz, dx * dx);
        const float k = fg[j] * __builtin_amdgcn_rsqf(__builtin_fmaf(r2, r2, vc4));
        fu = __builtin_fmaf(dz, k, fu);
        fw = __builtin_fmaf(dx, k, fw);
      }
    }
  }
}

}  // namespace ludvm
