// O(N) helpers around the spatial order (spatial_order.hpp): how compact are the origin classes of an array in a given
// order, and the gathers / scatters that apply an order.  See order.hip, spatial_order_if_needed (the only unit that includes this header).
#pragma once
#include "pair_kernels.hpp"

namespace ludvm {

// Per origin block `blockIdx.x`: ext[5 b] = sum over its two origin classes (256-element block x index parity) of the class's
// extent (xmax - xmin) + (zmax - zmin), elements taken in the order `order` (nullptr: as stored); ext[5 b + 1 .. 4] = the
// block's bounding box xmin, xmax, zmin, zmax.  Non-finite coordinates are left out.  One workgroup of 256 per block; fixed
// reduction tree.
__global__ void __launch_bounds__(kOriginBlock)
class_extents(const double* x, const double* z, const unsigned* order, long long n, double* ext) {
  const long long i = (long long)blockIdx.x * kOriginBlock + threadIdx.x;
  double lo_x = 1e300, hi_x = -1e300, lo_z = 1e300, hi_z = -1e300;
  if (i < n) {
    const long long p = order ? (long long)order[i] : i;
    const double vx = x[p], vz = z[p];
    if (fabs(vx) < 1e300 && fabs(vz) < 1e300) { lo_x = hi_x = vx; lo_z = hi_z = vz; }
  }
  // strides 2 .. 32 keep the lane parity: lanes 0 / 1 of each wavefront end up with the even / odd class's bounds
  for (int s = 2; s < 64; s <<= 1) {
    lo_x = fmin(lo_x, __shfl_xor(lo_x, s)); hi_x = fmax(hi_x, __shfl_xor(hi_x, s));
    lo_z = fmin(lo_z, __shfl_xor(lo_z, s)); hi_z = fmax(hi_z, __shfl_xor(hi_z, s));
  }
  __shared__ double red[kOriginBlock / 64][2][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane < 2) { red[wv][lane][0] = lo_x; red[wv][lane][1] = hi_x; red[wv][lane][2] = lo_z; red[wv][lane][3] = hi_z; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double e = 0.0, bx0 = 1e300, bx1 = -1e300, bz0 = 1e300, bz1 = -1e300;
    for (int p = 0; p < 2; ++p) {
      double a = 1e300, b = -1e300, c = 1e300, d = -1e300;
      for (int w = 0; w < kOriginBlock / 64; ++w) {
        a = fmin(a, red[w][p][0]); b = fmax(b, red[w][p][1]); c = fmin(c, red[w][p][2]); d = fmax(d, red[w][p][3]);
      }
      if (b >= a) e += (b - a) + (d - c);
      bx0 = fmin(bx0, a); bx1 = fmax(bx1, b); bz0 = fmin(bz0, c); bz1 = fmax(bz1, d);
    }
    double* o = ext + 5 * (long long)blockIdx.x;
    o[0] = e; o[1] = bx0; o[2] = bx1; o[3] = bz0; o[4] = bz1;
  }
}

// out[0] = sum of the blocks' extents (index order within each of 256 strided lanes, then a fixed tree: the same bits every
// time); out[1 .. 4] = the bounding box of all blocks
__global__ void __launch_bounds__(256) reduce_extents(const double* ext, long long nblk, double* out) {
  double s = 0.0, x0 = 1e300, x1 = -1e300, z0 = 1e300, z1 = -1e300;
  for (long long i = threadIdx.x; i < nblk; i += 256) {
    const double* e = ext + 5 * i;
    s += e[0];
    x0 = fmin(x0, e[1]); x1 = fmax(x1, e[2]); z0 = fmin(z0, e[3]); z1 = fmax(z1, e[4]);
  }
  __shared__ double red[256][5];
  red[threadIdx.x][0] = s; red[threadIdx.x][1] = x0; red[threadIdx.x][2] = x1; red[threadIdx.x][3] = z0; red[threadIdx.x][4] = z1;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) {
      double* a = red[threadIdx.x];
      const double* b = red[threadIdx.x + h];
      a[0] += b[0]; a[1] = fmin(a[1], b[1]); a[2] = fmax(a[2], b[2]); a[3] = fmin(a[3], b[3]); a[4] = fmax(a[4], b[4]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) out[threadIdx.x] = red[0][threadIdx.x];
}

// dst[a][k] = src[a][order[k]] for up to three arrays of n doubles (order == nullptr: a copy)
__global__ void __launch_bounds__(kBlock)
gather_f64(const double* s0, const double* s1, const double* s2, const unsigned* order, long long n, double* d0, double* d1,
           double* d2) {
  const long long k = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const long long p = order ? (long long)order[k] : k;
  if (s0) d0[k] = s0[p];
  if (s1) d1[k] = s1[p];
  if (s2) d2[k] = s2[p];
}

// dst[a][order[k]] = src[a][k]: results computed in the spatial order go back to the caller's
__global__ void __launch_bounds__(kBlock)
scatter_f64(const double* s0, const double* s1, const unsigned* order, long long n, double* d0, double* d1) {
  const long long k = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const long long p = (long long)order[k];
  d0[p] = s0[k];
  d1[p] = s1[k];
}

}  // namespace ludvm
