// Vatistas-core (n=2) Biot-Savart all-pairs kernels for gfx950 (MI355X, CDNA4).
//
// Replaces the O(Np*Nw) NumPy broadcast in LUDVM.induced_velocity
// (reference LUDVM.py:549-570):
//     u_p =  sum_w G_w (z_p - z_w) / (2 pi sqrt(r^4 + vc^4))
//     w_p = -sum_w G_w (x_p - x_w) / (2 pi sqrt(r^4 + vc^4)),   r^2 = dx^2 + dz^2
//
// Design:
//  * one wavefront lane owns TPL targets; the source range is walked in tiles staged in LDS as
//    SoA (x[], z[], G[]); every lane reads the same LDS address (broadcast ds_read_b128);
//  * the fp32 arithmetic is packed over SOURCES: one v_pk_* instruction advances two pairs of a
//    lane (sources j and j+1 against the same target), so targets are duplicated into register
//    pairs once, outside the loop, and no op_sel broadcast is needed inside it;
//  * per pair: sub, sub, mul, fma, fma, rsq, mul, fma, fma  (13 FLOP, FMA = 2, rsq = 1);
//    the 1/(2 pi) factor and the sign of w are applied once per target in the epilogue;
//  * grid = (target tiles) x (source splits); with more than one split each block writes its
//    partial (u, w) to a workspace slab and a second, tiny kernel sums the slabs in split order
//    (bitwise reproducible; no float atomics) and applies the fused epilogue (Euler update).
//
// No MFMA: the kernel is pairwise and non-linear in (p, w); there is no contraction to feed a
// matrix core with.
#pragma once
#include <hip/hip_runtime.h>

namespace ludvm {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;  // threads per workgroup = 4 wavefronts, one per SIMD
// Local origins: fp32 positions are stored as offsets from the origin of their origin CLASS: the vortices
// of one 256-vortex block (block b = indices [256 b, 256 b + 256)) that share an index parity -- class (b, p) holds
// indices 256 b + p, 256 b + p + 2, ...; its origin is the fp32-rounded position of its middle member, stored at
// cx / cz[2 b + p].  A pair difference is then (offset_p + (origin_p - origin_w)) - offset_w, with the origin
// difference added to the targets once per class pair.  The wake is stored in shedding order: while only trailing-edge
// vortices are shed a block is one compact stretch of the sheet, and while a leading-edge vortex is shed with every
// trailing-edge one the two families alternate -- a chord apart -- and each takes one index parity, so a CLASS is
// compact in both orders (round 2 kept one origin per block: 8e-6 of max|u| on a single sheet but 2-4e-5 on the
// alternating order, which is most of BASELINE config 2).  Pairs in nearby classes keep (almost) full relative
// precision at any distance from the coordinate origin, far ones do not need it.  Costs nothing in the pair loop
// (unlike hi+lo positions: +4 packed ops per two pairs): the two sources of a packed instruction have the two
// parities, so the packed target operand simply holds the target referred to one origin in each half.
// Measured [MI355X] on a wake at |x| ~ 50 with vortices 1e-3 apart (v_core = 1.3e-3): 1.4e-3 of max|u| with plain
// fp32 coordinates, 1.7e-5 with 512-vortex blocks whose origin is their first vortex, < 1e-5 with 256-vortex
// blocks whose origin is their middle vortex, 1e-6 with hi+lo positions.
constexpr int kOriginShift = 8;
constexpr int kOriginBlock = 1 << kOriginShift;   // = the 256-vortex tile of pair_sym_f32<4>
constexpr int kFinBlock = kOriginBlock;           // Euler finishers: one workgroup per origin block
// number of origin records (2 per block) an array of n positions needs, with one spare block for tiles that straddle
// the end
inline __host__ __device__ long long origin_slots(long long n) { return 2 * (n / kOriginBlock + 2); }
constexpr double kInv2PiD = 0.15915494309189533576888;
// Padding for source slots past the end of the range: far enough that r^4 overflows to +inf, so
// rsq() returns exactly 0 and the (zero-circulation) slot contributes exactly 0 even when vc = 0
// (inviscid), yet small enough that dx^2 + dz^2 itself stays finite.
constexpr float kPadPosF = 3.0e18f;
constexpr double kPadPosD = 1.0e150;

// Where a launch takes its targets from and where its results go.
struct PairArgs {
  // sources (device).  f32: xs, zs, gs (float).  f32x2: + xsl, zsl (lo parts).  f64: double arrays.
  const void* xs; const void* zs; const void* gs;
  const float* xsl; const float* zsl;
  long long ns;
  // targets: arrays (same typing as sources) or a uniform grid generated in registers
  const void* xt; const void* zt;
  const float* xtl; const float* ztl;
  long long nt;
  long long grid_nz;          // > 0: target p = (xmin + (grid_row0 + p / nz) * dr, zmin + (p % nz) * dr)
  long long grid_row0;        //   first grid row of this launch (a block of rows of a larger grid: same coordinates, bit for bit)
  double xmin, zmin, dr;
  // results: direct (nsplit == 1) or partial slabs [split][2][nt_pad] (nsplit > 1)
  void* u; void* w;
  void* part;
  long long nt_pad;
  long long chunk;            // sources per split (a multiple of the LDS tile)
  double vc4;                 // v_core^4 (0 when inviscid)
  // Device-resident time march: the wake size is decided on the device (LEV shedding), so a launch may take
  // it from memory: ns += *n_dev when ns_dev, nt += *n_dev when nt_dev.  The grid, `chunk` and `nt_pad` come
  // from the host's upper bound; surplus blocks find nothing to do.
  const long long* n_dev;
  int ns_dev, nt_dev;
  // LOCAL kernels: xs / zs are offsets from the origin of their origin class (256-source block x index parity),
  // scx / scz the origins (record 2 (source index >> kOriginShift) + (source index & 1); the source arrays start on a
  // block boundary).  Array targets are offsets too, their origins are tcx / tcz[2 ((t_index0 + p) >> kOriginShift)
  // + ((t_index0 + p) & 1)]; grid targets are generated in float64 and referred to the source classes' origins directly.
  const float* scx; const float* scz;
  const float* tcx; const float* tcz;
  long long t_index0;
  int nsplit;                 // number of source splits of the launch (rows of the partial slab)
};

struct PairSizes { long long ns, nt; };
__device__ __forceinline__ PairSizes pair_sizes(const PairArgs& a) {
  PairSizes r{a.ns, a.nt};
  if (a.n_dev) {
    const long long nd = *a.n_dev;
    if (a.ns_dev) r.ns += nd;
    if (a.nt_dev) r.nt += nd;
  }
  return r;
}

__device__ __forceinline__ void grid_point(const PairArgs& a, long long p, double& x, double& z) {
  const long long i = p / a.grid_nz;
  const long long j = p - i * a.grid_nz;
  x = a.xmin + (double)(a.grid_row0 + i) * a.dr;
  z = a.zmin + (double)j * a.dr;
}

// ---------------------------------------------------------------------------------------------
// fp32, packed over sources.  TPL = targets per lane, TILE = sources per LDS tile.
// HILO = false: plain fp32 positions.
// HILO = true : positions are hi+lo fp32 pairs, dx = (xh_p - xh_w) + (xl_p - xl_w); everything
//               after the difference is plain fp32 (removes the cancellation error of
//               |x| ~ 50 against a vortex spacing of ~1e-3).
// ---------------------------------------------------------------------------------------------
// GRID = 1 (flow-field grids with nz % TPL == 0): a lane owns TPL CONSECUTIVE grid points of one
//               row (same x, z stepping by dr), so dx, dx^2 and Gamma dx are computed once per source pair and
//               shared by the lane's targets.
// GRID = 2 (nz % 4 == 0): a lane owns a PATCH of kRows x 4 grid points (TPL = 8: 2 rows, TPL = 16: 4 rows), so dx, dx^2
//               and Gamma dx are shared along each row and dz, Gamma dz along each column: 4 + (3 kRows + 8) / TPL
//               packed ops per two pairs and target (5.25 at 4 x 4, 5.75 at 2 x 4; 6.5 for GRID = 1, 8 for GRID = 0).
//               The same operations on the same operands as GRID = 1, so the results are bit for bit the same.
// LOCAL = true: positions are offsets from block origins (see kOriginShift); the targets are re-referred to the
//               origin of each 256-source segment of the LDS tile (TPL adds per 256 sources).
template <int TPL, int TILE, bool HILO, int GRID = 0, bool LOCAL = false>
__global__ void __launch_bounds__(kBlock)
pair_f32(PairArgs a) {
  static_assert(TILE % kBlock == 0, "tile must be a multiple of the block size");
  static_assert(!(GRID && HILO), "the grid variants are plain fp32");
  static_assert(GRID != 2 || TPL % 2 == 0, "a patch is 2 rows");
  static_assert(!(LOCAL && HILO), "local origins replace hi+lo positions");
  static_assert(!LOCAL || TILE <= kOriginBlock || TILE % kOriginBlock == 0, "tile must be whole origin blocks");
  __shared__ __attribute__((aligned(16))) float lx[TILE];
  __shared__ __attribute__((aligned(16))) float lz[TILE];
  __shared__ __attribute__((aligned(16))) float lg[TILE];
  __shared__ __attribute__((aligned(16))) float lxl[HILO ? TILE : 4];
  __shared__ __attribute__((aligned(16))) float lzl[HILO ? TILE : 4];
  constexpr int kSegs = LOCAL ? (TILE + kOriginBlock - 1) / kOriginBlock : 1;
  __shared__ float lox[2 * kSegs], loz[2 * kSegs];      // LOCAL: origins of the tile's source classes (block x parity), staged with the tile

  const float* __restrict__ xs = static_cast<const float*>(a.xs);
  const float* __restrict__ zs = static_cast<const float*>(a.zs);
  const float* __restrict__ gs = static_cast<const float*>(a.gs);

  const int tid = threadIdx.x;
  const PairSizes sz = pair_sizes(a);
  // a launch sized from an upper bound of the target count (march): blocks without targets leave at once
  // (uniform over the block, before any barrier)
  // targets of this lane: tid, tid + 256, ... of the block's slab (GRID = 0); TPL consecutive points of a grid row
  // (GRID = 1); or a patch of kRows rows x 4 columns (GRID = 2: patches are numbered along the row groups)
  constexpr int kRows = GRID == 2 ? (TPL >= 16 ? 4 : 2) : 1;
  constexpr int kCols = GRID == 2 ? TPL / kRows : 1;
  const long long lane_id = (long long)blockIdx.x * kBlock + tid;
  const long long nzq = GRID == 2 ? a.grid_nz / kCols : 1;                   // patches per row pair
  const long long prow = GRID == 2 ? lane_id / nzq : 0, pcol = GRID == 2 ? lane_id - prow * nzq : 0;
  auto tindex = [&](int t) -> long long {
    if (GRID == 2) return (kRows * prow + t / kCols) * a.grid_nz + pcol * kCols + t % kCols;   // beyond the grid: >= nt
    if (GRID == 1) return lane_id * TPL + t;
    return (long long)blockIdx.x * kBlock * TPL + tid + (long long)t * kBlock;
  };
  if (GRID == 2) {
    const long long nrows = sz.nt / a.grid_nz;
    if ((long long)blockIdx.x * kBlock >= ((nrows + kRows - 1) / kRows) * nzq) return;
  } else if ((long long)blockIdx.x * kBlock * TPL >= sz.nt) {
    return;
  }
  const long long s_begin = (long long)blockIdx.y * a.chunk;
  long long s_end = s_begin + a.chunk;
  if (s_end > sz.ns) s_end = sz.ns;

  f32x2 xp[TPL], zp[TPL], xpl[TPL], zpl[TPL];
  float au[TPL], aw[TPL];      // running totals (the two packed halves of a tile's sums are added when the tile is folded in)
  // LOCAL: what the targets are re-referred from -- exact grid coordinates, or offsets + origins
  double gxd[LOCAL ? TPL : 1], gzd[LOCAL ? TPL : 1];
  float tox[LOCAL ? TPL : 1], toz[LOCAL ? TPL : 1], tlx[LOCAL ? TPL : 1], tlz[LOCAL ? TPL : 1];
#pragma unroll
  for (int t = 0; t < TPL; ++t) {
    const long long ti = tindex(t);
    float x = 0.0f, z = 0.0f, xl = 0.0f, zl = 0.0f;
    if (LOCAL) { gxd[t] = 0.0; gzd[t] = 0.0; tox[t] = 0.0f; toz[t] = 0.0f; tlx[t] = 0.0f; tlz[t] = 0.0f; }
    if (ti < sz.nt) {
      if (GRID != 0 || a.grid_nz > 0) {
        double xd, zd;
        grid_point(a, ti, xd, zd);
        x = (float)xd;
        z = (float)zd;
        if (HILO) { xl = (float)(xd - (double)x); zl = (float)(zd - (double)z); }
        if (LOCAL) { gxd[t] = xd; gzd[t] = zd; }
      } else if (LOCAL && GRID == 0) {
        const long long tg = a.t_index0 + ti, tb = 2 * (tg >> kOriginShift) + (tg & 1);
        tlx[t] = static_cast<const float*>(a.xt)[ti];
        tlz[t] = static_cast<const float*>(a.zt)[ti];
        tox[t] = a.tcx[tb];
        toz[t] = a.tcz[tb];
      } else {
        x = static_cast<const float*>(a.xt)[ti];
        z = static_cast<const float*>(a.zt)[ti];
        if (HILO) { xl = a.xtl[ti]; zl = a.ztl[ti]; }
      }
    }
    xp[t] = (f32x2){x, x};
    zp[t] = (f32x2){z, z};
    xpl[t] = (f32x2){xl, xl};
    zpl[t] = (f32x2){zl, zl};
    au[t] = 0.0f;
    aw[t] = 0.0f;
  }
  const float vc4s = (float)a.vc4;
  const f32x2 vc4 = {vc4s, vc4s};

  for (long long base = s_begin; base < s_end; base += TILE) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int k = 0; k < TILE / kBlock; ++k) {
      const int l = tid + k * kBlock;
      const long long si = base + l;
      const bool ok = si < s_end;
      lx[l] = ok ? xs[si] : kPadPosF;
      lz[l] = ok ? zs[si] : kPadPosF;
      lg[l] = ok ? gs[si] : 0.0f;
      if (HILO) {
        lxl[l] = ok ? a.xsl[si] : 0.0f;
        lzl[l] = ok ? a.zsl[si] : 0.0f;
      }
    }
    if (LOCAL && tid < 2 * kSegs && base + (long long)(tid >> 1) * kOriginBlock < s_end) {
      lox[tid] = a.scx[2 * (base >> kOriginShift) + tid];
      loz[tid] = a.scz[2 * (base >> kOriginShift) + tid];
    }
    __syncthreads();

    // two-level sum: a tile's 1024 contributions are summed on their own and then added to the running
    // total, so the fp32 rounding error grows with tile + N/tile terms instead of N (a 1e6-source
    // flow-field launch with one split: 1.6e-5 -> 2e-6 of max|u|)
    f32x2 tu[TPL], tw[TPL];
#pragma unroll
    for (int t = 0; t < TPL; ++t) { tu[t] = (f32x2){0.0f, 0.0f}; tw[t] = (f32x2){0.0f, 0.0f}; }

    constexpr int kSeg = LOCAL ? (TILE < kOriginBlock ? TILE : kOriginBlock) : TILE;
    for (int seg = 0; seg < TILE; seg += kSeg) {
    if (LOCAL) {
      // the segment's sources have two origins, one per index parity -- and a packed instruction pairs an even source
      // with the odd one behind it: the packed target operand holds the target referred to the even sources' origin in
      // its low half and to the odd sources' origin in its high half
      const long long sidx = base + seg;
      if (sidx >= s_end) break;
      const int so = 2 * (seg >> kOriginShift);
      const float ox0 = lox[so], ox1 = lox[so + 1], oz0 = loz[so], oz1 = loz[so + 1];
#pragma unroll
      for (int t = 0; t < TPL; ++t) {
        // (grid kernels read x of one target per row and z of one per column only)
        if (GRID == 1 && t != 0) { zp[t] = (f32x2){(float)(gzd[t] - (double)oz0), (float)(gzd[t] - (double)oz1)}; continue; }
        if (GRID == 2 && t >= kCols && t % kCols != 0) continue;
        if (GRID != 0 || a.grid_nz > 0) {
          xp[t] = (f32x2){(float)(gxd[t] - (double)ox0), (float)(gxd[t] - (double)ox1)};
          zp[t] = (f32x2){(float)(gzd[t] - (double)oz0), (float)(gzd[t] - (double)oz1)};
        } else {
          xp[t] = (f32x2){tlx[t] + (tox[t] - ox0), tlx[t] + (tox[t] - ox1)};
          zp[t] = (f32x2){tlz[t] + (toz[t] - oz0), tlz[t] + (toz[t] - oz1)};
        }
      }
    }
#pragma unroll 2
    for (int j = seg; j < seg + kSeg; j += 4) {
      const f32x4 X = *reinterpret_cast<const f32x4*>(&lx[j]);
      const f32x4 Z = *reinterpret_cast<const f32x4*>(&lz[j]);
      const f32x4 G = *reinterpret_cast<const f32x4*>(&lg[j]);
      f32x4 XL, ZL;
      if (HILO) {
        XL = *reinterpret_cast<const f32x4*>(&lxl[j]);
        ZL = *reinterpret_cast<const f32x4*>(&lzl[j]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 xs2 = h ? (f32x2){X.z, X.w} : (f32x2){X.x, X.y};
        const f32x2 zs2 = h ? (f32x2){Z.z, Z.w} : (f32x2){Z.x, Z.y};
        const f32x2 gs2 = h ? (f32x2){G.z, G.w} : (f32x2){G.x, G.y};
        f32x2 xl2, zl2;
        if (HILO) {
          xl2 = h ? (f32x2){XL.z, XL.w} : (f32x2){XL.x, XL.y};
          zl2 = h ? (f32x2){ZL.z, ZL.w} : (f32x2){ZL.x, ZL.y};
        }
        // The grid kernels weight the DIFFERENCES with the source strengths (u += (g dz) s, w += (g dx) s) instead of the
        // kernel value (s g): g dx is shared along a row and g dz along a column, like dx, dx^2 and dz themselves.
        if (GRID == 1) {
          const f32x2 dx = xp[0] - xs2;      // every target of the lane sits in the same grid row
          const f32x2 dxx = dx * dx;
          const f32x2 gdx = gs2 * dx;
#pragma unroll
          for (int t = 0; t < TPL; ++t) {
            const f32x2 dz = zp[t] - zs2;
            const f32x2 gdz = gs2 * dz;
            const f32x2 r2 = __builtin_elementwise_fma(dz, dz, dxx);
            const f32x2 q = __builtin_elementwise_fma(r2, r2, vc4);
            const f32x2 s = {__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)};
            tu[t] = __builtin_elementwise_fma(gdz, s, tu[t]);
            tw[t] = __builtin_elementwise_fma(gdx, s, tw[t]);
          }
          continue;
        }
        if (GRID == 2) {
          // kRows rows share each column's dz and g dz, kCols columns share each row's dx, dx^2 and g dx
          f32x2 gdxr[kRows], dxxr[kRows], dzc[kCols], gdzc[kCols];
#pragma unroll
          for (int r = 0; r < kRows; ++r) { const f32x2 d = xp[r * kCols] - xs2; dxxr[r] = d * d; gdxr[r] = gs2 * d; }
#pragma unroll
          for (int cc = 0; cc < kCols; ++cc) { dzc[cc] = zp[cc] - zs2; gdzc[cc] = gs2 * dzc[cc]; }
#pragma unroll
          for (int t = 0; t < TPL; ++t) {
            const int r = t / kCols, cc = t % kCols;
            const f32x2 r2 = __builtin_elementwise_fma(dzc[cc], dzc[cc], dxxr[r]);
            const f32x2 q = __builtin_elementwise_fma(r2, r2, vc4);
            const f32x2 s = {__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)};
            tu[t] = __builtin_elementwise_fma(gdzc[cc], s, tu[t]);
            tw[t] = __builtin_elementwise_fma(gdxr[r], s, tw[t]);
          }
          continue;
        }
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
          f32x2 dx = xp[t] - xs2;
          f32x2 dz = zp[t] - zs2;
          if (HILO) {
            dx = dx + (xpl[t] - xl2);
            dz = dz + (zpl[t] - zl2);
          }
          f32x2 r2 = dx * dx;
          r2 = __builtin_elementwise_fma(dz, dz, r2);
          const f32x2 q = __builtin_elementwise_fma(r2, r2, vc4);
          f32x2 s = {__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)};
          s = s * gs2;
          tu[t] = __builtin_elementwise_fma(dz, s, tu[t]);
          tw[t] = __builtin_elementwise_fma(dx, s, tw[t]);
        }
      }
    }
    }
#pragma unroll
    for (int t = 0; t < TPL; ++t) { au[t] += tu[t].x + tu[t].y; aw[t] += tw[t].x + tw[t].y; }
  }

  const float scale = (float)kInv2PiD;
#pragma unroll
  for (int t = 0; t < TPL; ++t) {
    const long long ti = tindex(t);
    if (ti < sz.nt) {
      const float uu = au[t] * scale;
      const float ww = -aw[t] * scale;
      if (gridDim.y == 1) {
        static_cast<float*>(a.u)[ti] = uu;
        static_cast<float*>(a.w)[ti] = ww;
      } else {
        float* row = static_cast<float*>(a.part) + (long long)blockIdx.y * 2 * a.nt_pad;
        row[ti] = uu;
        row[a.nt_pad + ti] = ww;
      }
    }
  }
}

// 1/sqrt(q) in fp64: v_rsq_f64 (~2^-26 relative) refined by two Newton steps y <- y (1.5 - 0.5 q y^2),
// each squaring the error (-> below 1 ulp); ~10 fp64 ops instead of the ~60 of a divide plus a sqrt.
// q = 0 (inviscid self pair) gives +inf like the exact form, so G * rsq still reproduces the
// reference's NaN / inf semantics at coincident points; q = +inf gives 0.
__device__ __forceinline__ double rsqrt_f64(double q) {
  double y = __builtin_amdgcn_rsq(q);
  const double h = 0.5 * q;
  // the Newton update is only meaningful for finite, non-zero q (inf * 0 would turn the limits into NaN)
  if (q > 0.0 && q < 1.0e300) {
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
  }
  return y;
}

// ---------------------------------------------------------------------------------------------
// fp64 throughout (parity / debug mode and the small chord-target calls).  One target per lane.
// TILE = 512 for many targets (throughput); TILE = 128 for a few (the 80 chord points of a time step:
// the launch is latency-bound, so each workgroup walks a short tile and there are 4x as many of them).
// ---------------------------------------------------------------------------------------------
template <int TILE>
__global__ void __launch_bounds__(kBlock)
pair_f64(PairArgs a) {
  __shared__ __attribute__((aligned(16))) double lx[TILE];
  __shared__ __attribute__((aligned(16))) double lz[TILE];
  __shared__ __attribute__((aligned(16))) double lg[TILE];

  const double* __restrict__ xs = static_cast<const double*>(a.xs);
  const double* __restrict__ zs = static_cast<const double*>(a.zs);
  const double* __restrict__ gs = static_cast<const double*>(a.gs);

  const int tid = threadIdx.x;
  const PairSizes sz = pair_sizes(a);
  if ((long long)blockIdx.x * kBlock >= sz.nt) return;   // no targets in this block (launch sized from a bound)
  const long long ti = (long long)blockIdx.x * kBlock + tid;
  const long long s_begin = (long long)blockIdx.y * a.chunk;
  long long s_end = s_begin + a.chunk;
  if (s_end > sz.ns) s_end = sz.ns;

  double xp = 0.0, zp = 0.0;
  if (ti < sz.nt) {
    if (a.grid_nz > 0) {
      grid_point(a, ti, xp, zp);
    } else {
      xp = static_cast<const double*>(a.xt)[ti];
      zp = static_cast<const double*>(a.zt)[ti];
    }
  }
  double au = 0.0, aw = 0.0;
  const double vc4 = a.vc4;

  for (long long base = s_begin; base < s_end; base += TILE) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (TILE + kBlock - 1) / kBlock; ++k) {
      const int l = tid + k * kBlock;
      if (l < TILE) {
        const long long si = base + l;
        const bool ok = si < s_end;
        lx[l] = ok ? xs[si] : kPadPosD;
        lz[l] = ok ? zs[si] : kPadPosD;
        lg[l] = ok ? gs[si] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < TILE; ++j) {
      const double dx = xp - lx[j];
      const double dz = zp - lz[j];
      const double r2 = __builtin_fma(dz, dz, dx * dx);
      const double q = __builtin_fma(r2, r2, vc4);
      const double s = lg[j] * rsqrt_f64(q);
      au = __builtin_fma(dz, s, au);
      aw = __builtin_fma(dx, s, aw);
    }
  }

  if (ti < sz.nt) {
    const double uu = au * kInv2PiD;
    const double ww = -aw * kInv2PiD;
    if (gridDim.y == 1) {
      static_cast<double*>(a.u)[ti] = uu;
      static_cast<double*>(a.w)[ti] = ww;
    } else {
      double* row = static_cast<double*>(a.part) + (long long)blockIdx.y * 2 * a.nt_pad;
      row[ti] = uu;
      row[a.nt_pad + ti] = ww;
    }
  }
}

// The same sums for FEW targets (the 80 chord points of a time step, plus three in the march: nt <= 128), where one
// target per lane leaves most of a 256-lane workgroup idle -- and an idle lane still walks its tile: lane l serves
// target l % nt of source split  blockIdx.y * groups + l / nt, so a workgroup works `groups` = 256 / nt splits at once
// (3 for 83 targets) and the launch needs a third of the workgroups.  Every (split, target) partial sum is formed by
// one lane walking that split's sources in order, exactly as pair_f64 does: the slab -- and every bit of the result --
// is the same.  In the overlapped march this launch runs beside the symmetric kernel and its waves take issue slots
// from it: 2096 wave-walks per step at 67 000 vortices become 700.
constexpr int kFewGroupsMax = 4;
template <int TILE>
__global__ void __launch_bounds__(kBlock)
pair_f64_few(PairArgs a) {
  // A wavefront that straddles two groups (83 targets: lanes 64 .. 82 of wave 1 belong to group 0, the rest to group 1)
  // reads the same j of two groups' tiles in one instruction; TILE doubles apart they share a bank (1024 B = 4 x 64 banks):
  // SQ_LDS_BANK_CONFLICT was 10 % of this kernel's busy cycles in config 2 [MI355X, profiles/r04_config2_sizes_pmc_sq.csv].
  // Two doubles of padding per group move the groups 4 banks apart.  (Addresses only: the same bits.)
#ifndef LUDVM_FEW_PAD
#define LUDVM_FEW_PAD 2
#endif
  constexpr int kPad = LUDVM_FEW_PAD;
  // few waves with a dependent launch waiting for them (the march's solve chain runs them beside a roll-up kernel that
  // keeps every SIMD's issue slots busy): they go first
  __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(16))) double lx[kFewGroupsMax][TILE + kPad];
  __shared__ __attribute__((aligned(16))) double lz[kFewGroupsMax][TILE + kPad];
  __shared__ __attribute__((aligned(16))) double lg[kFewGroupsMax][TILE + kPad];
  const double* __restrict__ xs = static_cast<const double*>(a.xs);
  const double* __restrict__ zs = static_cast<const double*>(a.zs);
  const double* __restrict__ gs = static_cast<const double*>(a.gs);
  const int tid = threadIdx.x;
  const PairSizes sz = pair_sizes(a);
  const int nt = (int)sz.nt;                               // 1 .. kBlock / 2
  int groups = kBlock / nt;
  if (groups > kFewGroupsMax) groups = kFewGroupsMax;
  const int g = tid / nt, p = tid - g * nt;
  const long long split0 = (long long)blockIdx.y * groups;
  const long long split = split0 + g;
  const bool mine = g < groups && split < a.nsplit;
  double xp = 0.0, zp = 0.0;
  if (mine) {
    xp = static_cast<const double*>(a.xt)[p];
    zp = static_cast<const double*>(a.zt)[p];
  }
  double au = 0.0, aw = 0.0;
  const double vc4 = a.vc4;
  for (long long off = 0; off < a.chunk; off += TILE) {
    __syncthreads();
    for (int l = tid; l < groups * TILE; l += kBlock) {
      const int gg = l / TILE, idx = l - gg * TILE;
      const long long s_begin = (split0 + gg) * a.chunk;
      long long s_end = s_begin + a.chunk;
      if (s_end > sz.ns) s_end = sz.ns;
      const long long si = s_begin + off + idx;
      const bool ok = si < s_end;
      lx[gg][idx] = ok ? xs[si] : kPadPosD;
      lz[gg][idx] = ok ? zs[si] : kPadPosD;
      lg[gg][idx] = ok ? gs[si] : 0.0;
    }
    __syncthreads();
    if (mine) {
#pragma unroll 4
      for (int j = 0; j < TILE; ++j) {
        const double dx = xp - lx[g][j];
        const double dz = zp - lz[g][j];
        const double r2 = __builtin_fma(dz, dz, dx * dx);
        const double q = __builtin_fma(r2, r2, vc4);
        const double sv = lg[g][j] * rsqrt_f64(q);
        au = __builtin_fma(dz, sv, au);
        aw = __builtin_fma(dx, sv, aw);
      }
    }
  }
  if (mine) {
    double* row = static_cast<double*>(a.part) + split * 2 * a.nt_pad;
    row[p] = au * kInv2PiD;
    row[a.nt_pad + p] = -aw * kInv2PiD;
  }
}

// ---------------------------------------------------------------------------------------------
// Finishers: sum the per-split partial slabs in split order (deterministic), then the epilogue.
// ---------------------------------------------------------------------------------------------
// Four independent running sums (splits s, s+1, s+2, s+3 mod 4), combined at the end in a fixed order:
// deterministic, and the loads of a group of four are in flight together instead of one after another.
template <typename T>
__device__ __forceinline__ T sum_column(const T* col, long long stride, int nsplit) {
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    a0 += col[(long long)s * stride];
    a1 += col[(long long)(s + 1) * stride];
    a2 += col[(long long)(s + 2) * stride];
    a3 += col[(long long)(s + 3) * stride];
  }
  for (; s < nsplit; ++s) a0 += col[(long long)s * stride];
  return (a0 + a1) + (a2 + a3);
}

template <typename T>
__device__ __forceinline__ void sum_splits(const T* part, long long i, long long nt_pad, int nsplit, T& su, T& sw) {
  su = sum_column(part + i, 2 * nt_pad, nsplit);
  sw = sum_column(part + nt_pad + i, 2 * nt_pad, nsplit);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
finish_sum(const T* part, long long nt, long long nt_pad, int nsplit, T* u, T* w) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nt) return;
  T su, sw;
  sum_splits(part, i, nt_pad, nsplit, su, sw);
  u[i] = su;
  w[i] = sw;
}

// Split a float64 master position into the fp32 (hi, lo) pair the kernels read.
__device__ __forceinline__ void split_hilo(double v, float& hi, float& lo) {
  hi = (float)v;
  lo = (float)(v - (double)hi);
}

// The fp32 mirrors of the resident wake's float64 master positions: (xh, xl) / (zh, zl) hi+lo pairs for the
// f32x2 kernels, (xr, zr) offsets from the class origins (cx, cz)[2 (index >> 8) + (index & 1)] for the local-origin
// fp32 kernels.  Invariant: xr[i] = (float)(x64[i] - cx[origin_slot(i)]) for every stored vortex; cx[2 b + p] is the fp32
// position a vortex of class (b, p) had when the block's offsets were last rewritten (the middle one, or the newest one
// of a class that is not yet half full).  Both records of a block that holds any vortex are finite numbers.
struct Mirrors {
  float* xh; float* xl; float* zh; float* zl;
  float* xr; float* zr; float* cx; float* cz;
};

// origin record of vortex i
__host__ __device__ __forceinline__ long long origin_slot(long long i) { return 2 * (i >> kOriginShift) + (i & 1); }

// Which vortex lends origin class (block b, index parity p) its origin when n vortices are stored (n > 256 b): the
// middle one of the class, or its newest while the class is less than half full; a class without a member yet (the
// odd one of a block that holds a single vortex) borrows the block's first vortex, so that its record is a number.
__host__ __device__ __forceinline__ long long origin_index(long long b, int p, long long n) {
  const long long first = b << kOriginShift;
  const long long mid = first + kOriginBlock / 2 + p;
  if (mid < n) return mid;
  const long long last = (n - 1) - (((n - 1) ^ p) & 1);       // the newest stored index of parity p
  return last >= first ? last : first;
}

// Euler finishers (one workgroup per origin block): the threads that have just moved the two origin vortices of the
// block publish the new origins -- org = [x even, x odd, z even, z odd] in LDS for the block's threads, and the records.
__device__ __forceinline__ void publish_origins(const Mirrors& m, long long i, long long n, double xn, double zn, float* org) {
  const long long b = blockIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p)
    if (i == origin_index(b, p, n)) {
      org[p] = (float)xn; org[2 + p] = (float)zn;
      m.cx[2 * b + p] = org[p]; m.cz[2 * b + p] = org[2 + p];
    }
}

__device__ __forceinline__ void store_mirrors(const Mirrors& m, long long i, double x, double z, float ox, float oz) {
  split_hilo(x, m.xh[i], m.xl[i]);
  split_hilo(z, m.zh[i], m.zl[i]);
  m.xr[i] = (float)(x - (double)ox);
  m.zr[i] = (float)(z - (double)oz);
}

// Device-resident march: what the Euler finisher leaves for the coming time step (march_kernels.hpp), done by the
// threads that have the data in their hands.  The thread that has just moved the newest TEV / LEV places the
// next ones one third of the way from the shedding edge (LUDVM.py:680-681, :797-800); block 0 copies the coming
// step's chord points next to them: tgt = x[npan + 3] | z[npan + 3] are the targets of the fp64 wake -> chord
// launch (chord points, TEV placement, LEV placement, the origin).  kin_next = the coming step's kinematics row
// [alpha, alpha_dot, h_dot, te_x, te_z, le_x, le_z, xg[npan], zg[npan]]; null S = not a march.
struct TailDuty {
  double* place;          // tev_x, lev_x, tev_z, lev_z of the coming step
  double* tgt;
  long long* n_old;       // wake size after this roll-up
  const int* tail;        // vortices shed by the step being rolled up (1 or 2)
  const int* shed;        // ... and whether a LEV was among them
  const double* kin_next;
  int npan;
  // dense history: every vortex's position after this roll-up goes to hist_row = x[hist_nmax] | z[hist_nmax]
  // (the reference's path[...][i] rows, LUDVM.py:1108-1127); null = not recorded
  double* hist_row;
  long long hist_nmax;
};

__device__ __forceinline__ void tail_duty(const TailDuty& td, long long i, long long n, double xn, double zn) {
  if (td.hist_row) { td.hist_row[i] = xn; td.hist_row[td.hist_nmax + i] = zn; }
  if (td.kin_next == nullptr) return;
  const int np = td.npan, ntt = np + 3;
  const int tail = *td.tail;
  if (i == n - tail) {
    const double tex = td.kin_next[3], tez = td.kin_next[4];
    const double px = tex + (xn - tex) / 3, pz = tez + (zn - tez) / 3;
    td.place[0] = px; td.place[2] = pz;
    td.tgt[np] = px; td.tgt[ntt + np] = pz;
  }
  if (i == n - 1) {
    const double lex = td.kin_next[5], lez = td.kin_next[6];
    double px = lex, pz = lez;
    if (*td.shed && tail == 2) { px = lex + (xn - lex) / 3; pz = lez + (zn - lez) / 3; }
    td.place[1] = px; td.place[3] = pz;
    td.tgt[np + 1] = px; td.tgt[ntt + np + 1] = pz;
  }
}

// The part of the duty that does not depend on a vortex: called by every thread of the finisher BEFORE the
// threads without a vortex leave (a young wake is smaller than the number of chord points).
__device__ __forceinline__ void tail_duty_block0(const TailDuty& td, long long n) {
  if (td.kin_next == nullptr || blockIdx.x != 0) return;
  const int np = td.npan, ntt = np + 3;
  for (int t = threadIdx.x; t < np; t += blockDim.x) {
    td.tgt[t] = td.kin_next[7 + t];
    td.tgt[ntt + t] = td.kin_next[7 + np + t];
  }
  if (threadIdx.x == 0) { *td.n_old = n; td.tgt[np + 2] = 0.0; td.tgt[ntt + np + 2] = 0.0; }
}

// Resident-wake Euler step (LUDVM.py:1108-1127): float64 update of the master copy from the summed
// partials (T = float for the fp32 kernels, double for the fp64 one), refresh of the fp32 mirrors.
// One workgroup of kFinBlock threads = one origin block: its middle vortex's new position becomes the block's
// origin, so the local offsets stay small however far the wake drifts.
template <typename T>
__global__ void __launch_bounds__(kFinBlock)
finish_wake_advect(const T* part, long long nt, long long nt_pad, int nsplit, double dt, double* x64, double* z64,
                   Mirrors m, double* u_out, double* w_out, const long long* n_dev = nullptr, TailDuty td = TailDuty{}) {
  __shared__ float org[4];
  const long long i = (long long)blockIdx.x * kFinBlock + threadIdx.x;
  if (n_dev) nt = *n_dev;          // device-resident march: the wake size lives on the device
  tail_duty_block0(td, nt);
  const bool on = i < nt;
  double xn = 0.0, zn = 0.0;
  if (on) {
    T su, sw;
    sum_splits(part, i, nt_pad, nsplit, su, sw);
    if (u_out) { u_out[i] = (double)su; w_out[i] = (double)sw; }
    xn = x64[i] + dt * (double)su;
    zn = z64[i] + dt * (double)sw;
    publish_origins(m, i, nt, xn, zn, org);
  }
  __syncthreads();
  if (!on) return;
  x64[i] = xn;
  z64[i] = zn;
  store_mirrors(m, i, xn, zn, org[i & 1], org[2 + (i & 1)]);
  tail_duty(td, i, nt, xn, zn);
}

}  // namespace ludvm
