// Semantics check of the packed-fp32 operand selects on gfx950: one register pair holds TWO different targets and
// op_sel / op_sel_hi broadcast either half to both lanes of a v_pk_* instruction (instead of keeping every target twice).
//   hipcc --offload-arch=gfx950 -O3 -o opsel_check opsel_check.hip && ./opsel_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// d = {p.lo, p.lo} - s     /   d = {p.hi, p.hi} - s
__device__ __forceinline__ f32x2 sub_lo(f32x2 p, f32x2 s) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(s));
  return d;
}
__device__ __forceinline__ f32x2 sub_hi(f32x2 p, f32x2 s) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(s));
  return d;
}
// m = s * {p.lo, p.lo}    /   m = s * {p.hi, p.hi}
__device__ __forceinline__ f32x2 mul_lo(f32x2 s, f32x2 p) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(s), "v"(p));
  return d;
}
__device__ __forceinline__ f32x2 mul_hi(f32x2 s, f32x2 p) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(s), "v"(p));
  return d;
}

__global__ void k(const f32x2* p, const f32x2* s, f32x2* out) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  out[4 * i + 0] = sub_lo(p[i], s[i]);
  out[4 * i + 1] = sub_hi(p[i], s[i]);
  out[4 * i + 2] = mul_lo(s[i], p[i]);
  out[4 * i + 3] = mul_hi(s[i], p[i]);
}

int main() {
  const int n = 4096;
  f32x2 *hp = new f32x2[n], *hs = new f32x2[n], *ho = new f32x2[4 * n];
  for (int i = 0; i < n; ++i) { hp[i] = (f32x2){(float)drand48() - 0.5f, (float)drand48() * 3.f}; hs[i] = (f32x2){(float)drand48(), -(float)drand48()}; }
  f32x2 *dp, *ds, *dout;
  auto ok = [](hipError_t e) { if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); exit(2); } };
  ok(hipMalloc(&dp, n * 8)); ok(hipMalloc(&ds, n * 8)); ok(hipMalloc(&dout, 4 * n * 8));
  ok(hipMemcpy(dp, hp, n * 8, hipMemcpyHostToDevice)); ok(hipMemcpy(ds, hs, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, ds, dout);
  ok(hipMemcpy(ho, dout, 4 * n * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const f32x2 p = hp[i], s = hs[i];
    const f32x2 e0 = {p.x - s.x, p.x - s.y}, e1 = {p.y - s.x, p.y - s.y}, e2 = {s.x * p.x, s.y * p.x}, e3 = {s.x * p.y, s.y * p.y};
    const f32x2 exp[4] = {e0, e1, e2, e3};
    for (int q = 0; q < 4; ++q)
      if (ho[4 * i + q].x != exp[q].x || ho[4 * i + q].y != exp[q].y) { if (bad < 5) printf("mismatch i=%d q=%d got (%g,%g) want (%g,%g)\n", i, q, ho[4*i+q].x, ho[4*i+q].y, exp[q].x, exp[q].y); ++bad; }
  }
  printf(bad ? "OPSEL_BAD %d\n" : "OPSEL_OK\n", bad);
  return bad != 0;
}
