// VALU issue-rate probe: plain v_fma_f32 / v_mul / v_add vs packed v_pk_fma_f32 / v_pk_mul / v_pk_add.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
  const float c = 1.0001f; const f2 c2 = {1.0001f, 0.9999f};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7) F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
    } else if (MODE == 1) {
#define P(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c2));
      P(p0) P(p1) P(p2) P(p3) P(p4) P(p5) P(p6) P(p7) P(p0) P(p1) P(p2) P(p3) P(p4) P(p5) P(p6) P(p7)
    } else if (MODE == 2) {
#define M(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
      M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
    } else if (MODE == 3) {
#define PM(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
      PM(p0) PM(p1) PM(p2) PM(p3) PM(p4) PM(p5) PM(p6) PM(p7) PM(p0) PM(p1) PM(p2) PM(p3) PM(p4) PM(p5) PM(p6) PM(p7)
    } else if (MODE == 4) {
#define C(x, y) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
      C(a0, a1) C(a2, a3) C(a4, a5) C(a6, a7) C(a0, a1) C(a2, a3) C(a4, a5) C(a6, a7)
    } else if (MODE == 5) {
#define E(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
      E(a0) E(a1) E(a2) E(a3) E(a4) E(a5) E(a6) E(a7) E(a0) E(a1) E(a2) E(a3) E(a4) E(a5) E(a6) E(a7)
    } else if (MODE == 6) {
#define S(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
      S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE> void run(const char *name, float *out, int blocks, double lane_ops_per_instr) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  k<MODE><<<blocks, 256>>>(out, iters, 1.f);
  hipEventRecord(e0, 0); k<MODE><<<blocks, 256>>>(out, iters, 1.f); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * 4 * iters * 16;  // wave-instructions
  const double cyc_per_instr_per_simd = (ms * 1e-3 * 2.4e9) / (instr / 1024.0);
  printf("%-14s blocks=%5d  %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)  %.1f T lane-ops/s\n", name, blocks, ms,
         cyc_per_instr_per_simd, instr * 64 * lane_ops_per_instr / (ms * 1e-3) / 1e12);
}
int main() {
  float *out; hipMalloc(&out, 8192 * 256 * 4);
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0>("v_fma_f32", out, blocks, 1); run<1>("v_pk_fma_f32", out, blocks, 2); run<2>("v_mul_f32", out, blocks, 1);
    run<3>("v_pk_mul_f32", out, blocks, 2); run<4>("cmp+cndmask", out, blocks, 1); run<5>("v_exp_f32", out, blocks, 1); run<6>("v_sqrt_f32", out, blocks, 1);
  }
  return 0;
}
