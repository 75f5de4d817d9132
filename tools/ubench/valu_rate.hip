// VALU issue-rate micro-benchmark for gfx950 (MI355X).
// Measures cycles per wave-instruction of the ops the Biot-Savart pair kernel is
// made of, at 1/2/4/8 waves per SIMD, so the kernel design (packed vs scalar fp32,
// cost of v_rsq_f32) rests on measured numbers rather than on datasheet peaks.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;

// each variant: body executed ITERS times, REP instances of the op per body
#define REP8(x) x x x x x x x x

template <int V>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f;
  float b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3};
  f2 p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
  f2 c = {1.0001f, 0.9999f};
  float cs = 1.0001f;
  for (int i = 0; i < ITERS; ++i) {
    if constexpr (V == 0) {  // 8 independent v_fma_f32
      asm volatile(
        "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
        "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
    } else if constexpr (V == 1) {  // 8 independent v_pk_fma_f32
      asm volatile(
        "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
        "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
    } else if constexpr (V == 2) {  // 8 independent v_rsq_f32
      asm volatile(
        "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
        "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (V == 3) {  // 8 v_pk_mul_f32
      asm volatile(
        "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
    } else if constexpr (V == 4) {  // 8 v_pk_add_f32
      asm volatile(
        "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
    } else if constexpr (V == 5) {  // pair-kernel mix, scalar fp32: 8 full-rate + 1 rsq (one pair per lane), x2 chains
      asm volatile(
        "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
        "v_rsq_f32 %4, %4\n"
        "v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n v_fma_f32 %0, %0, %8, %8\n"
        "v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %5, %5, %8, %8\n"
        "v_rsq_f32 %4, %4\n"
        "v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
    } else if constexpr (V == 6) {  // pair-kernel mix, packed: 8 pk + 2 rsq (two pairs per lane)
      asm volatile(
        "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
        "v_rsq_f32 %9, %9\n"
        "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n"
        "v_rsq_f32 %10, %10\n"
        "v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(a0), "v"(a1));
    } else if constexpr (V == 7) {  // 8 v_mul_f32
      asm volatile(
        "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
    } else if constexpr (V == 8) {  // 4 fma + 4 rsq interleaved (does rsq co-issue with full-rate VALU?)
      asm volatile(
        "v_fma_f32 %0, %0, %8, %8\n v_rsq_f32 %4, %4\n v_fma_f32 %1, %1, %8, %8\n v_rsq_f32 %5, %5\n"
        "v_fma_f32 %2, %2, %8, %8\n v_rsq_f32 %6, %6\n v_fma_f32 %3, %3, %8, %8\n v_rsq_f32 %7, %7\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
    } else if constexpr (V == 9) {  // 8 fma with an SGPR operand
      asm volatile(
        "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
        "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
            p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  if (r == 123.456f) out[0] = r;
}

struct Var { const char* name; int ops_per_body; void (*fn)(float*, float); };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, clk_khz);
  float* out; CK(hipMalloc(&out, 4));
  Var vars[] = {
    {"v_fma_f32 x8", 8, k<0>}, {"v_pk_fma_f32 x8", 8, k<1>}, {"v_rsq_f32 x8", 8, k<2>},
    {"v_pk_mul_f32 x8", 8, k<3>}, {"v_pk_add_f32 x8", 8, k<4>}, {"mix 16fma+2rsq (2 pairs)", 18, k<5>},
    {"mix 8pk+2rsq (2 pairs)", 10, k<6>}, {"v_mul_f32 x8", 8, k<7>}, {"4fma+4rsq interleaved", 8, k<8>},
    {"v_fma_f32 sgpr x8", 8, k<9>},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int cus = prop.multiProcessorCount;
  for (auto& v : vars) {
    for (int wps : {1, 2, 4, 8}) {          // waves per SIMD
      int blocks = cus * wps;                // 256 threads = 4 waves = 1 wave per SIMD per block
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      // wave-instructions per SIMD = wps * ITERS * ops; cycles = ms * clk
      double wave_instr = (double)wps * ITERS * v.ops_per_body;
      double cyc = best * 1e-3 * (double)clk_khz * 1e3;
      printf("%-28s waves/SIMD %d  %.3f ms  %.2f cyc/wave-instr (at nominal clock)  %.2f cyc/body/wave-slot\n",
             v.name, wps, best, cyc / wave_instr, cyc / ((double)wps * ITERS));
    }
  }
  return 0;
}
