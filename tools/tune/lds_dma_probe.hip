// global_load_lds_dwordx4 on gfx950: where does lane p's 16 bytes land, and does the m0 base reach beyond 64 KB of the 160 KB LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4 *src, uint4 *out, uint32_t base) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  uint4 *l = reinterpret_cast<uint4 *>(lds);
  for (int i = threadIdx.x; i < 160 * 1024 / 16 - 64; i += blockDim.x) l[i] = make_uint4(0xdeadbeefu, 0, 0, 0);
  __syncthreads();
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char *)lds;
  const uint32_t voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_waitcnt vmcnt(0)" ::"s"(__builtin_amdgcn_readfirstlane((int)(lds0 + base))), "v"(voff), "s"(src) : "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[0] = make_uint4(lds0, 0, 0, 0);
  // report where the 64 chunks went: scan LDS
  for (int i = threadIdx.x; i < 160 * 1024 / 16 - 64; i += blockDim.x) {
    const uint4 v = l[i];
    if (v.x != 0xdeadbeefu) out[1 + v.x] = make_uint4((uint32_t)i * 16u, v.x, v.y, 0);
  }
}
int main() {
  std::vector<uint4> h(64);
  for (int i = 0; i < 64; ++i) h[i] = make_uint4(i, 100 + i, 0, 0);
  uint4 *src, *out;
  (void)hipMalloc(&src, 64 * 16); (void)hipMalloc(&out, 65 * 16);
  (void)hipMemcpy(src, h.data(), 64 * 16, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  for (uint32_t base : {0u, 4096u, 65536u, 98304u + 2048u, 147456u}) {
    (void)hipMemset(out, 0xff, 65 * 16);
    k<<<1, 64, 160 * 1024 - 1024>>>(src, out, base);
    std::vector<uint4> r(65);
    hipError_t e = hipMemcpy(r.data(), out, 65 * 16, hipMemcpyDeviceToHost);
    printf("base %6u (err %d, lds0 %u): lane 0 -> %u, lane 1 -> %u, lane 2 -> %u, lane 63 -> %u\n", base, (int)e, r[0].x, r[1].x, r[2].x, r[3].x, r[64].x);
  }
  return 0;
}
