// O(N) kernels of the flow field (flowfield.hip): source staging conversions and the vorticity stencil of LUDVM.flowfield
// (LUDVM.py:1224-1292).  Non-template kernels: this header belongs to ONE translation unit (flowfield.hip).
#pragma once
#include "pair_kernels.hpp"
#include "pair_sym_kernels.hpp"

namespace ludvm {

// float64 -> float32 staging conversion for the stateless host API.
__global__ void __launch_bounds__(kBlock)
cvt_f64_to_f32(const double* in, float* hi, float* lo, long long n) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float h, l;
  split_hilo(in[i], h, l);
  hi[i] = h;
  if (lo) lo[i] = l;
}

__global__ void __launch_bounds__(kBlock)
cvt_f32_to_f64(const float* in, double* out, long long n) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}

// float64 -> local-origin fp32 for the stateless host API: off[i] = (float)(in[i] - org[origin_slot(i)]) with
// org[2 b + p] = (float)in[origin_index(b, p, n)].  n is rounded up to whole blocks by the grid: the threads of a
// class without a member still write its record (a number: the kernels read both records of every block they stage).
__global__ void __launch_bounds__(kBlock)
cvt_f64_to_local(const double* in, float* off, float* org, long long n) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long b = i >> kOriginShift;
  if ((b << kOriginShift) >= n) return;
  const int p = (int)(i & 1);
  const float o = (float)in[origin_index(b, p, n)];
  if ((i & (kOriginBlock - 1)) < 2) org[2 * b + p] = o;
  if (i < n) off[i] = (float)(in[i] - (double)o);
}

// ---------------------------------------------------------------------------------------------
// Vorticity stencil of LUDVM.flowfield (LUDVM.py:1224-1292): ome = dw/dx - du/dz on the uniform
// grid, centred in the interior, one-sided on edges and corners.  u, w, ome are [nx][nz], z fastest.
// HBM-bound (reads ~4 neighbours from L2, writes one float).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
vorticity_f32(const float* u, const float* w, long long nx, long long nz, float dr, float* ome) {
  const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (p >= nx * nz) return;
  const long long i = p / nz, j = p - i * nz;
  const long long ip = i + 1 < nx ? i + 1 : i, im = i > 0 ? i - 1 : i;
  const long long jp = j + 1 < nz ? j + 1 : j, jm = j > 0 ? j - 1 : j;
  // the reference takes dx, dz from the mesh itself: (ip - im) * dr, (jp - jm) * dr
  const float dxm = (float)(ip - im) * dr;
  const float dzm = (float)(jp - jm) * dr;
  const float dw = w[ip * nz + j] - w[im * nz + j];
  const float du = u[i * nz + jp] - u[i * nz + jm];
  ome[p] = dw / dxm - du / dzm;
}

// The same stencil in float64 on rows [row0, row0 + nx) of the grid, with the mesh differences taken from the mesh
// values themselves as the reference does (x[i+1, j] - x[i-1, j] with x = xmin + i dr, LUDVM.py:1193, :1228-1229): the
// float64 flow field then equals the reference's to rounding.
__global__ void __launch_bounds__(kBlock)
vorticity_f64(const double* u, const double* w, long long nx, long long nz, long long row0, double xmin, double zmin, double dr,
              double* ome) {
  const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (p >= nx * nz) return;
  const long long i = p / nz, j = p - i * nz;
  const long long ip = i + 1 < nx ? i + 1 : i, im = i > 0 ? i - 1 : i;
  const long long jp = j + 1 < nz ? j + 1 : j, jm = j > 0 ? j - 1 : j;
  const double dxm = (xmin + (double)(row0 + ip) * dr) - (xmin + (double)(row0 + im) * dr);
  const double dzm = (zmin + (double)jp * dr) - (zmin + (double)jm * dr);
  const double dw = w[ip * nz + j] - w[im * nz + j];
  const double du = u[i * nz + jp] - u[i * nz + jm];
  ome[p] = dw / dxm - du / dzm;
}

}  // namespace ludvm
