// Throughput of lane-rotation primitives on gfx950 (event-timed, 8 waves/SIMD, long loops).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 32768;
template <int V>
__global__ void __launch_bounds__(256) k(float* out) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < ITERS; ++i) {
#define R8(OP) asm volatile(OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
    if constexpr (V == 0) {
#define OP(r) "v_mov_b32_dpp " r ", " r " wave_rol:1 row_mask:0xf bank_mask:0xf\n"
      R8(OP);
#undef OP
    } else if constexpr (V == 1) {
#define OP(r) "v_mov_b32_dpp " r ", " r " row_ror:1 row_mask:0xf bank_mask:0xf\n"
      R8(OP);
#undef OP
    } else if constexpr (V == 2) {
#define OP(r) "v_mul_f32 " r ", 1.0, " r "\n"
      R8(OP);
#undef OP
    } else if constexpr (V == 3) {
#define OP(r) "v_mov_b32_dpp " r ", " r " wave_ror:1 row_mask:0xf bank_mask:0xf\n"
      R8(OP);
#undef OP
    } else if constexpr (V == 4) {
#define OP(r) "v_add_f32_dpp " r ", " r ", " r " wave_rol:1 row_mask:0xf bank_mask:0xf\n"
      R8(OP);
#undef OP
    } else if constexpr (V == 5) {
#define OP(r) "v_mov_b32_dpp " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
      R8(OP);
#undef OP
    } else if constexpr (V == 6) {
#define OP(r) "v_mov_b32_dpp " r ", " r " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
      R8(OP);
#undef OP
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (r == 123.456f) out[0] = r;
}
struct Var { const char* name; void (*fn)(float*); };
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  float* out; CK(hipMalloc(&out, 4));
  Var vars[] = {{"v_mov_b32_dpp wave_rol:1", k<0>}, {"v_mov_b32_dpp row_ror:1", k<1>}, {"v_mul_f32 (ref)", k<2>},
                {"v_mov_b32_dpp wave_ror:1", k<3>}, {"v_add_f32_dpp wave_rol:1", k<4>}, {"v_mov_b32_dpp row_shr:1", k<5>},
                {"v_mov_b32_dpp quad_perm", k<6>}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& v : vars) for (int wps : {4, 8}) {
    int blocks = prop.multiProcessorCount * wps;
    hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double ns_per = best * 1e6 / ((double)wps * ITERS * 8);
    printf("%-28s waves/SIMD %d  %.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cyc at 2.27 GHz; v_mul ref = 2 cyc)\n", v.name, wps, best, ns_per, ns_per * 2.27);
  }
  return 0;
}
