// VGPR bank-conflict probe for gfx950: same instruction, explicit register numbers.
// Build: hipcc -O3 --offload-arch=gfx950 vgpr_bank.hip -o vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int ITERS = 8192;

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
  "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"

#define INIT \
  "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 0.5\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 0.5\n v_mov_b32 v15, 0.5\n" \
  "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" \
  "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n" \
  "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n" \
  "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"

template <int V>
__global__ void __launch_bounds__(256) k(unsigned long long* stamps) {
  unsigned long long t0, t1;
  asm volatile(INIT ::: CLOB);
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < ITERS; ++i) {
    if constexpr (V == 0)  // fma: D,A,B all bank 0
      asm volatile("v_fma_f32 v16, v8, v12, v16\n v_fma_f32 v20, v8, v12, v20\n v_fma_f32 v24, v8, v12, v24\n v_fma_f32 v28, v8, v12, v28\n"
                   "v_fma_f32 v32, v8, v12, v32\n v_fma_f32 v36, v8, v12, v36\n v_fma_f32 v40, v8, v12, v40\n v_fma_f32 v44, v8, v12, v44\n" ::: CLOB);
    else if constexpr (V == 1)  // fma: A bank0, B bank1, D bank2
      asm volatile("v_fma_f32 v18, v8, v13, v18\n v_fma_f32 v22, v8, v13, v22\n v_fma_f32 v26, v8, v13, v26\n v_fma_f32 v30, v8, v13, v30\n"
                   "v_fma_f32 v34, v8, v13, v34\n v_fma_f32 v38, v8, v13, v38\n v_fma_f32 v42, v8, v13, v42\n v_fma_f32 v46, v8, v13, v46\n" ::: CLOB);
    else if constexpr (V == 2)  // fma: A bank0, B bank0, D bank2
      asm volatile("v_fma_f32 v18, v8, v12, v18\n v_fma_f32 v22, v8, v12, v22\n v_fma_f32 v26, v8, v12, v26\n v_fma_f32 v30, v8, v12, v30\n"
                   "v_fma_f32 v34, v8, v12, v34\n v_fma_f32 v38, v8, v12, v38\n v_fma_f32 v42, v8, v12, v42\n v_fma_f32 v46, v8, v12, v46\n" ::: CLOB);
    else if constexpr (V == 3)  // fma: A bank0, B bank1, D bank0
      asm volatile("v_fma_f32 v16, v8, v13, v16\n v_fma_f32 v20, v8, v13, v20\n v_fma_f32 v24, v8, v13, v24\n v_fma_f32 v28, v8, v13, v28\n"
                   "v_fma_f32 v32, v8, v13, v32\n v_fma_f32 v36, v8, v13, v36\n v_fma_f32 v40, v8, v13, v40\n v_fma_f32 v44, v8, v13, v44\n" ::: CLOB);
    else if constexpr (V == 4)  // mul: D=A*B A bank0 B bank0
      asm volatile("v_mul_f32 v16, v8, v12\n v_mul_f32 v20, v8, v12\n v_mul_f32 v24, v8, v12\n v_mul_f32 v28, v8, v12\n"
                   "v_mul_f32 v32, v8, v12\n v_mul_f32 v36, v8, v12\n v_mul_f32 v40, v8, v12\n v_mul_f32 v44, v8, v12\n" ::: CLOB);
    else if constexpr (V == 5)  // mul: A bank0 B bank1
      asm volatile("v_mul_f32 v16, v8, v13\n v_mul_f32 v20, v8, v13\n v_mul_f32 v24, v8, v13\n v_mul_f32 v28, v8, v13\n"
                   "v_mul_f32 v32, v8, v13\n v_mul_f32 v36, v8, v13\n v_mul_f32 v40, v8, v13\n v_mul_f32 v44, v8, v13\n" ::: CLOB);
    else if constexpr (V == 6)  // pk_fma: A=[8:9] B=[12:13] D=[16:17].. all banks {0,1}
      asm volatile("v_pk_fma_f32 v[16:17], v[8:9], v[12:13], v[16:17]\n v_pk_fma_f32 v[20:21], v[8:9], v[12:13], v[20:21]\n"
                   "v_pk_fma_f32 v[24:25], v[8:9], v[12:13], v[24:25]\n v_pk_fma_f32 v[28:29], v[8:9], v[12:13], v[28:29]\n"
                   "v_pk_fma_f32 v[32:33], v[8:9], v[12:13], v[32:33]\n v_pk_fma_f32 v[36:37], v[8:9], v[12:13], v[36:37]\n"
                   "v_pk_fma_f32 v[40:41], v[8:9], v[12:13], v[40:41]\n v_pk_fma_f32 v[44:45], v[8:9], v[12:13], v[44:45]\n" ::: CLOB);
    else if constexpr (V == 7)  // pk_fma: A={0,1} B={2,3} D={0,1}
      asm volatile("v_pk_fma_f32 v[16:17], v[8:9], v[14:15], v[16:17]\n v_pk_fma_f32 v[20:21], v[8:9], v[14:15], v[20:21]\n"
                   "v_pk_fma_f32 v[24:25], v[8:9], v[14:15], v[24:25]\n v_pk_fma_f32 v[28:29], v[8:9], v[14:15], v[28:29]\n"
                   "v_pk_fma_f32 v[32:33], v[8:9], v[14:15], v[32:33]\n v_pk_fma_f32 v[36:37], v[8:9], v[14:15], v[36:37]\n"
                   "v_pk_fma_f32 v[40:41], v[8:9], v[14:15], v[40:41]\n v_pk_fma_f32 v[44:45], v[8:9], v[14:15], v[44:45]\n" ::: CLOB);
    else if constexpr (V == 8)  // pk_fma: A={0,1} B={0,1} D={2,3}
      asm volatile("v_pk_fma_f32 v[18:19], v[8:9], v[12:13], v[18:19]\n v_pk_fma_f32 v[22:23], v[8:9], v[12:13], v[22:23]\n"
                   "v_pk_fma_f32 v[26:27], v[8:9], v[12:13], v[26:27]\n v_pk_fma_f32 v[30:31], v[8:9], v[12:13], v[30:31]\n"
                   "v_pk_fma_f32 v[34:35], v[8:9], v[12:13], v[34:35]\n v_pk_fma_f32 v[38:39], v[8:9], v[12:13], v[38:39]\n"
                   "v_pk_fma_f32 v[42:43], v[8:9], v[12:13], v[42:43]\n v_pk_fma_f32 v[46:47], v[8:9], v[12:13], v[46:47]\n" ::: CLOB);
    else if constexpr (V == 9)  // pk_fma with D=A*A+D (2 distinct operands): A={0,1}, D={2,3}
      asm volatile("v_pk_fma_f32 v[18:19], v[8:9], v[8:9], v[18:19]\n v_pk_fma_f32 v[22:23], v[8:9], v[8:9], v[22:23]\n"
                   "v_pk_fma_f32 v[26:27], v[8:9], v[8:9], v[26:27]\n v_pk_fma_f32 v[30:31], v[8:9], v[8:9], v[30:31]\n"
                   "v_pk_fma_f32 v[34:35], v[8:9], v[8:9], v[34:35]\n v_pk_fma_f32 v[38:39], v[8:9], v[8:9], v[38:39]\n"
                   "v_pk_fma_f32 v[42:43], v[8:9], v[8:9], v[42:43]\n v_pk_fma_f32 v[46:47], v[8:9], v[8:9], v[46:47]\n" ::: CLOB);
    else if constexpr (V == 10)  // pk_mul D=A*B: A={0,1} B={0,1}
      asm volatile("v_pk_mul_f32 v[16:17], v[8:9], v[12:13]\n v_pk_mul_f32 v[20:21], v[8:9], v[12:13]\n"
                   "v_pk_mul_f32 v[24:25], v[8:9], v[12:13]\n v_pk_mul_f32 v[28:29], v[8:9], v[12:13]\n"
                   "v_pk_mul_f32 v[32:33], v[8:9], v[12:13]\n v_pk_mul_f32 v[36:37], v[8:9], v[12:13]\n"
                   "v_pk_mul_f32 v[40:41], v[8:9], v[12:13]\n v_pk_mul_f32 v[44:45], v[8:9], v[12:13]\n" ::: CLOB);
    else if constexpr (V == 11)  // pk_mul: A={0,1} B={2,3}
      asm volatile("v_pk_mul_f32 v[16:17], v[8:9], v[14:15]\n v_pk_mul_f32 v[20:21], v[8:9], v[14:15]\n"
                   "v_pk_mul_f32 v[24:25], v[8:9], v[14:15]\n v_pk_mul_f32 v[28:29], v[8:9], v[14:15]\n"
                   "v_pk_mul_f32 v[32:33], v[8:9], v[14:15]\n v_pk_mul_f32 v[36:37], v[8:9], v[14:15]\n"
                   "v_pk_mul_f32 v[40:41], v[8:9], v[14:15]\n v_pk_mul_f32 v[44:45], v[8:9], v[14:15]\n" ::: CLOB);
    else if constexpr (V == 12)  // pk_mul D=A*A (single operand read)
      asm volatile("v_pk_mul_f32 v[16:17], v[8:9], v[8:9]\n v_pk_mul_f32 v[20:21], v[8:9], v[8:9]\n"
                   "v_pk_mul_f32 v[24:25], v[8:9], v[8:9]\n v_pk_mul_f32 v[28:29], v[8:9], v[8:9]\n"
                   "v_pk_mul_f32 v[32:33], v[8:9], v[8:9]\n v_pk_mul_f32 v[36:37], v[8:9], v[8:9]\n"
                   "v_pk_mul_f32 v[40:41], v[8:9], v[8:9]\n v_pk_mul_f32 v[44:45], v[8:9], v[8:9]\n" ::: CLOB);
    else if constexpr (V == 13)  // rsq
      asm volatile("v_rsq_f32 v16, v8\n v_rsq_f32 v20, v8\n v_rsq_f32 v24, v8\n v_rsq_f32 v28, v8\n"
                   "v_rsq_f32 v32, v8\n v_rsq_f32 v36, v8\n v_rsq_f32 v40, v8\n v_rsq_f32 v44, v8\n" ::: CLOB);
    else if constexpr (V == 14)  // pk_fma with op_sel broadcast of src1 low half
      asm volatile("v_pk_fma_f32 v[18:19], v[8:9], v[12:13], v[18:19] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[22:23], v[8:9], v[12:13], v[22:23] op_sel_hi:[1,0,1]\n"
                   "v_pk_fma_f32 v[26:27], v[8:9], v[12:13], v[26:27] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[30:31], v[8:9], v[12:13], v[30:31] op_sel_hi:[1,0,1]\n"
                   "v_pk_fma_f32 v[34:35], v[8:9], v[12:13], v[34:35] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[38:39], v[8:9], v[12:13], v[38:39] op_sel_hi:[1,0,1]\n"
                   "v_pk_fma_f32 v[42:43], v[8:9], v[12:13], v[42:43] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[46:47], v[8:9], v[12:13], v[46:47] op_sel_hi:[1,0,1]\n" ::: CLOB);
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
struct Var { const char* name; void (*fn)(unsigned long long*); };
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  unsigned long long* st; CK(hipMalloc(&st, 8 * cus * 8 * 4));
  Var vars[] = {
    {"fma A0 B0 D0", k<0>}, {"fma A0 B1 D2", k<1>}, {"fma A0 B0 D2", k<2>}, {"fma A0 B1 D0", k<3>},
    {"mul A0 B0", k<4>}, {"mul A0 B1", k<5>},
    {"pk_fma A01 B01 D01", k<6>}, {"pk_fma A01 B23 D01", k<7>}, {"pk_fma A01 B01 D23", k<8>}, {"pk_fma A01 A01 D23", k<9>},
    {"pk_mul A01 B01", k<10>}, {"pk_mul A01 B23", k<11>}, {"pk_mul A01 A01", k<12>}, {"rsq", k<13>}, {"pk_fma op_sel bcast", k<14>},
  };
  for (auto& v : vars) {
    printf("%-22s", v.name);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
      int blocks = cus * wps;
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, st); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, st); CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(blocks * 4);
      CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      double mc = (double)h[h.size() / 2];
      printf("  w%d: %.2f", wps, mc / ((double)ITERS * 8 * wps));
    }
    printf("   (cyc/instr/SIMD)\n");
  }
  return 0;
}
