// See spatial_order.hpp.  gfx950 only.
#include "spatial_order.hpp"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace ludvm {

namespace {

__device__ __forceinline__ unsigned spread16(unsigned v) {      // 16 bits -> every other bit of 32
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

__global__ void __launch_bounds__(256) morton_keys(const double* x, const double* z, unsigned n, OrderBox b, unsigned* keys,
                                                   unsigned* vals) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  // (NaN -> cell 0 through fmax; infinities are clamped)
  const double cx = fmin(fmax((x[i] - b.x0) * b.sx, 0.0), 65535.0);
  const double cz = fmin(fmax((z[i] - b.z0) * b.sz, 0.0), 65535.0);
  keys[i] = spread16((unsigned)cx) | (spread16((unsigned)cz) << 1);
  vals[i] = i;
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t rocprim_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr),
                                  static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr), n, 0, 32, hipStream_t(nullptr));
  return bytes;
}

}  // namespace

size_t spatial_order_temp_bytes(size_t n) { return 3 * align256(n * sizeof(unsigned)) + align256(rocprim_bytes(n)) + 256; }

hipError_t spatial_order_sort(const double* d_x, const double* d_z, size_t n, OrderBox box, void* d_tmp, size_t tmp_bytes,
                              unsigned* d_order, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if (n >= (size_t)1 << 32 || tmp_bytes < spatial_order_temp_bytes(n)) return hipErrorInvalidValue;
  char* p = static_cast<char*>(d_tmp);
  unsigned* keys = reinterpret_cast<unsigned*>(p); p += align256(n * sizeof(unsigned));
  unsigned* keys_out = reinterpret_cast<unsigned*>(p); p += align256(n * sizeof(unsigned));
  unsigned* vals = reinterpret_cast<unsigned*>(p); p += align256(n * sizeof(unsigned));
  size_t rp = rocprim_bytes(n);
  hipLaunchKernelGGL(morton_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_x, d_z, (unsigned)n, box, keys, vals);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return rocprim::radix_sort_pairs(p, rp, keys, keys_out, vals, d_order, n, 0, 32, stream);
}

}  // namespace ludvm
