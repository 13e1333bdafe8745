// split_pair (one value: v_cvt_f16_f32 + fmaf + v_cvt) against split4v (v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16) on random floats
// of every magnitude and sign: are the pieces the same bits?   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off split_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
constexpr float PM_SC = 2048.f;
__device__ __forceinline__ void split_pair(float v, _Float16 &hi, _Float16 &lo) {
  hi = (_Float16)v;
  lo = (_Float16)__builtin_fmaf((float)hi, -PM_SC, v * PM_SC);
}
__device__ __forceinline__ void split2(float a, float b, uint32_t &H, uint32_t &L) {
  const float nsc = -PM_SC;
  const float s0 = a * PM_SC, s1 = b * PM_SC;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H) : "v"(a), "v"(b));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(L) : "v"(H), "s"(nsc), "v"(s0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L) : "v"(H), "s"(nsc), "v"(s1));
}
__global__ void probe(const float *x, int n, uint32_t *out) {  // out[4i..]: hi/lo (pair), hi/lo (mix) of x[i]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  _Float16 h0, l0, h1, l1;
  split_pair(x[2 * i], h0, l0);
  split_pair(x[2 * i + 1], h1, l1);
  uint32_t H, L;
  split2(x[2 * i], x[2 * i + 1], H, L);
  out[4 * i] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
  out[4 * i + 1] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
  out[4 * i + 2] = H;
  out[4 * i + 3] = L;
}
int main() {
  const int n = 1 << 22;
  float *hx = (float *)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const int e = rand() % 60 - 40;  // 2^-40 .. 2^19
    float m = 1.f + (float)(rand() & 0x7fffff) / 8388608.f;
    hx[i] = ldexpf(m, e) * ((rand() & 1) ? 1.f : -1.f);
    if (i % 97 == 0) hx[i] = ldexpf((float)((rand() % 4096) - 2048) + 0.5f, e - 11);  // exact ties of the fp16 rounding
  }
  float *dx;
  uint32_t *dout, *hout = (uint32_t *)malloc(n * 2 * 4);
  hipMalloc(&dx, n * 4);
  hipMalloc(&dout, n * 2 * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  probe<<<n / 2 / 256, 256>>>(dx, n, dout);
  hipMemcpy(hout, dout, n * 2 * 4, hipMemcpyDeviceToHost);
  long bad_hi = 0, bad_lo = 0;
  int shown = 0;
  for (int i = 0; i < n / 2; ++i) {
    if (hout[4 * i] != hout[4 * i + 2]) ++bad_hi;
    if (hout[4 * i + 1] != hout[4 * i + 3]) {
      ++bad_lo;
      if (shown++ < 12) printf("x = %.9g, %.9g  hi %08x  lo pair %08x  lo mix %08x\n", hx[2 * i], hx[2 * i + 1], hout[4 * i], hout[4 * i + 1], hout[4 * i + 3]);
    }
  }
  printf("pairs %d: hi words differ %ld, lo words differ %ld\n", n / 2, bad_hi, bad_lo);
  return 0;
}
