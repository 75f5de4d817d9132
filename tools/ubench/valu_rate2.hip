// VALU issue-rate micro-benchmark v2 for gfx950: in-kernel cycle stamps (s_memtime = shader
// clock, s_memrealtime = 100 MHz) so cycles/instruction and the held clock are separated.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITERS = 8192;
typedef float f2 __attribute__((ext_vector_type(2)));

#define A8(OP) OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)

template <int V>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* stamps, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4};
  f2 p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
  f2 c = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
  float cs = 1.0001f, ds = 0.9999f;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  for (int i = 0; i < ITERS; ++i) {
#define VS(OPS) asm volatile(OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs), "v"(ds), "s"(seed))
#define VP(OPS) asm volatile(OPS : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(d), "v"(a0), "v"(a1))
    if constexpr (V == 0) {
#define OP(r) "v_mul_f32 " #r ", " #r ", %8\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 1) {
#define OP(r) "v_add_f32 " #r ", " #r ", %8\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 2) {
#define OP(r) "v_fmac_f32 " #r ", %8, %9\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 3) {
#define OP(r) "v_fma_f32 " #r ", %8, %9, " #r "\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 4) {
#define OP(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 5) {
#define OP(r) "v_mul_f32_e64 " #r ", " #r ", %8\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 6) {
#define OP(r) "v_fma_f32 " #r ", " #r ", " #r ", 1.0\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 7) {
#define OP(r) "v_pk_fma_f32 " #r ", " #r ", %8, %9\n"
      VP(A8(OP));
#undef OP
    } else if constexpr (V == 8) {
#define OP(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
      VP(A8(OP));
#undef OP
    } else if constexpr (V == 9) {
#define OP(r) "v_pk_add_f32 " #r ", " #r ", %8\n"
      VP(A8(OP));
#undef OP
    } else if constexpr (V == 10) {
#define OP(r) "v_rsq_f32 " #r ", " #r "\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 11) {
#define OP(r) "v_fmaak_f32 " #r ", " #r ", %8, 0x3f800000\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 12) {
#define OP(r) "v_fmac_f32 " #r ", %10, %9\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 13) {
#define OP(r) "v_sub_f32 " #r ", %10, " #r "\n"
      VS(A8(OP));
#undef OP
    } else if constexpr (V == 14) {  // candidate pair body A (VOP2-only + rsq), 1 pair: sub sub mul fmac mul(r2*r2) add(vc4) rsq mul fmac fmac
      asm volatile(
        "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_mul_f32 %2, %0, %0\n v_fmac_f32 %2, %1, %1\n"
        "v_fmaak_f32 %3, %2, %2, 0x3f800000\n v_rsq_f32 %3, %3\n v_mul_f32 %3, %3, %8\n v_fmac_f32 %6, %1, %3\n v_fmac_f32 %7, %0, %3\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs), "v"(ds));
    } else if constexpr (V == 15) {  // candidate body B packed (2 pairs): pk_add pk_add pk_mul pk_fma pk_fma rsq rsq pk_mul pk_fma pk_fma
      asm volatile(
        "v_pk_add_f32 %0, %8, %4\n v_pk_add_f32 %1, %9, %5\n v_pk_mul_f32 %2, %0, %0\n v_pk_fma_f32 %2, %1, %1, %2\n"
        "v_pk_fma_f32 %3, %2, %2, %9\n v_rsq_f32 %10, %10\n v_rsq_f32 %11, %11\n v_pk_mul_f32 %3, %3, %8\n"
        "v_pk_fma_f32 %6, %1, %3, %6\n v_pk_fma_f32 %7, %0, %3, %7\n"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(d), "v"(a0), "v"(a1));
    } else if constexpr (V == 16) {  // v_fma_f32 all distinct, accumulate form
#define OP(r) "v_fma_f32 " #r ", %8, %9, " #r "\n v_mul_f32 %8, %8, %9\n"
      asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %5, %6, %1\n v_fma_f32 %2, %6, %7, %2\n v_fma_f32 %3, %7, %4, %3\n"
                   "v_fma_f32 %0, %5, %4, %0\n v_fma_f32 %1, %6, %5, %1\n v_fma_f32 %2, %7, %6, %2\n v_fma_f32 %3, %4, %7, %3\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs), "v"(ds));
#undef OP
    }
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
            p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  if (r == 123.456f) out[0] = r;
  if ((threadIdx.x & 63) == 0) {
    int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
  }
}

struct Var { const char* name; int ops; void (*fn)(float*, unsigned long long*, float); };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  float* out; CK(hipMalloc(&out, 4));
  unsigned long long* st; CK(hipMalloc(&st, sizeof(unsigned long long) * 2 * cus * 8 * 4));
  Var vars[] = {
    {"v_mul_f32 (VOP2)", 8, k<0>}, {"v_add_f32 (VOP2)", 8, k<1>}, {"v_fmac_f32 (VOP2) D+=a*b", 8, k<2>},
    {"v_fma_f32 D=a*b+D", 8, k<3>}, {"v_fma_f32 D=D*a+b", 8, k<4>}, {"v_mul_f32_e64 (VOP3)", 8, k<5>},
    {"v_fma_f32 D=D*D+1.0", 8, k<6>}, {"v_pk_fma_f32", 8, k<7>}, {"v_pk_mul_f32", 8, k<8>}, {"v_pk_add_f32", 8, k<9>},
    {"v_rsq_f32", 8, k<10>}, {"v_fmaak_f32 (VOP2+lit)", 8, k<11>}, {"v_fmac_f32 sgpr", 8, k<12>}, {"v_sub_f32 sgpr", 8, k<13>},
    {"bodyA scalar 1 pair (9 instr)", 9, k<14>}, {"bodyB packed 2 pairs (10 instr)", 10, k<15>}, {"v_fma_f32 distinct regs", 8, k<16>},
  };
  for (auto& v : vars) {
    for (int wps : {1, 2, 4}) {
      int blocks = cus * wps;
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, st, 1.0f);
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, st, 1.0f);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(2 * blocks * 4);
      CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> cyc, rt;
      for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w]); rt.push_back((double)h[2 * w + 1]); }
      std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
      double mc = cyc[cyc.size() / 2], mr = rt[rt.size() / 2];
      double clk_ghz = mc / (mr * 10.0);  // realtime tick = 10 ns
      // cycles per wave-instruction on the SIMD = wave cycles / (ITERS*ops*wps)
      printf("%-34s waves/SIMD %d  wave-cycles %.0f  clock %.2f GHz  %.2f cyc/instr/SIMD  body %.1f cyc/SIMD\n",
             v.name, wps, mc, clk_ghz, mc / ((double)ITERS * v.ops * wps), mc / ((double)ITERS * wps));
    }
  }
  return 0;
}
