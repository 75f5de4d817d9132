// Spatial (Morton) order of a set of points -- what makes a 256-vortex origin block spatially compact when the caller's
// array is not a shed wake but an unordered cloud (LUDVM.generate_flowfield_turbulence, LUDVM.py:98-130; any user array
// handed to LUDVM.induced_velocity, :549-570, whose float64 sum does not care about the order).  The fp32 kernels store
// positions as offsets from the origin of their origin class (pair_kernels.hpp, kOriginShift): the offsets are small --
// and the pair differences exact to fp32 rounding of the DISTANCE, not of the coordinate -- only if a class is compact.
//
// The sort itself is rocPRIM's radix sort (an O(N) helper outside the pair kernels' time: 0.3 ms per 1e6 points); it lives
// in a translation unit of its own so that the kernels' file does not pay for the template instantiations.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace ludvm {

// key = Morton interleave of (clamp((x - x0) * sx, 0, 65535), clamp((z - z0) * sz, 0, 65535))
struct OrderBox {
  double x0, z0, sx, sz;
};

// device workspace spatial_order_sort needs for n points (keys in and out, the identity values, rocPRIM's temporaries)
size_t spatial_order_temp_bytes(size_t n);

// order[k] = index (in the caller's arrays) of the point that takes position k in Morton order; stable, deterministic.
// Asynchronous on `stream`.
hipError_t spatial_order_sort(const double* d_x, const double* d_z, size_t n, OrderBox box, void* d_tmp, size_t tmp_bytes,
                              unsigned* d_order, hipStream_t stream);

}  // namespace ludvm
