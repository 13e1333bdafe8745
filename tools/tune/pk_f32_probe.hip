// Does a packed f32 subtraction whose broadcast operand was written by the instruction right in front of it always see that write?
// (docs/kernels/round5.md section 8: fps_lean on packed instructions lost single updates of a running distance while another stream's
// kernels shared the GPU; both lanes caught were in 48-63, and in the ISA the packed instruction read v0 right behind the v_mov_b32
// that filled it.)  The probe issues that very pair -- v_mov_b32 vC, vZ ; v_pk_add_f32 vD[0:1], vP[0:1], vC[0:1] op_sel_hi:[1,0]
// neg_lo:[0,1] neg_hi:[0,1] -- with changing values and compares both halves with v_sub_f32, many times per wave, while other streams run
// an LDS-heavy and a memory-heavy kernel.  Prints the mismatches per lane quarter.  NOT RUN YET (written after round 5's GPU access had
// closed): hipcc -O3 --offload-arch=gfx950 tools/tune/pk_f32_probe.hip -o tools/tune/pk_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(const float *__restrict__ in, unsigned int *bad, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float z = in[t], p0 = in[t + 1], p1 = in[t + 2];
  unsigned int wrong = 0;
  for (int i = 0; i < iters; ++i) {
    f2 d;
    const f2 p = {p0, p1};
    // v200 receives z by a v_mov_b32 right in front of the packed instruction; v201 holds something else (never read: op_sel_hi)
    const float other = p1 * 3.f;
    asm volatile("v_mov_b32 v201, %3\n\t"
                 "v_mov_b32 v200, %2\n\t"
                 "v_pk_add_f32 %0, %1, v[200:201] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                 : "=v"(d)
                 : "v"(p), "v"(z), "v"(other)
                 : "v200", "v201");
    float e0, e1;  // the reference: one v_sub_f32 per half (the compiler would pack a plain `p - z` pair too)
    asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4" : "=&v"(e0), "=&v"(e1) : "v"(p0), "v"(p1), "v"(z));
    wrong += (__float_as_uint(d[0]) != __float_as_uint(e0)) + (__float_as_uint(d[1]) != __float_as_uint(e1));
    // new values every round (cheap, data dependent)
    z = z * 1.0009765625f + 0.25f;
    p0 = p0 * 0.99951171875f - 0.125f;
    p1 = e1 * 0.5f + p0;
    if (z > 1e6f) z = in[t] + (float)i * 1e-3f;
  }
  if (wrong) atomicAdd(&bad[(threadIdx.x & 63) >> 4], wrong);
}

__global__ __launch_bounds__(256) void lds_noise(float *out, int iters) {
  __shared__ float s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = (float)i;
  __syncthreads();
  float a = 0.f;
  for (int i = 0; i < iters; ++i) {
    a += s[(threadIdx.x * 33 + i * 7) & 8191];
    s[(threadIdx.x + i * 13) & 8191] = a;
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void mem_noise(const float4 *src, float4 *dst, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main() {
  const int blocks = 64, iters = 200000;
  std::vector<float> h((size_t)blocks * 512 + 4);
  unsigned int s = 1u;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 9) / 8388608.f - 0.5f; }
  float *in, *scratch; unsigned int *bad; float4 *a, *b;
  const size_t n4 = 16u << 20;
  (void)hipMalloc(&in, h.size() * 4); (void)hipMalloc(&bad, 16); (void)hipMalloc(&scratch, 4096 * 256 * 4);
  (void)hipMalloc(&a, n4 * 16); (void)hipMalloc(&b, n4 * 16);
  (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipStream_t s0, s1, s2;
  (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  for (int beside = 0; beside < 2; ++beside) {
    (void)hipMemset(bad, 0, 16);
    for (int rep = 0; rep < 20; ++rep) {
      if (beside) {
        lds_noise<<<1024, 256, 0, s1>>>(scratch, 2000);
        mem_noise<<<2048, 256, 0, s2>>>(a, b, n4);
      }
      probe<<<blocks, 512, 0, s0>>>(in, bad, iters);
    }
    (void)hipDeviceSynchronize();
    unsigned int r[4];
    (void)hipMemcpy(r, bad, 16, hipMemcpyDeviceToHost);
    printf("%s: mismatching halves in lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u (of %.3g packed instructions per quarter)\n",
           beside ? "beside an LDS-heavy and a memory-heavy kernel on two other streams" : "alone", r[0], r[1], r[2], r[3],
           20.0 * blocks * 8 * iters * 16);
  }
  return 0;
}
