// A SECOND hypothesis for round 5's lost update (docs/kernels/round6.md section 1), next to tools/tune/pk_f32_probe.hip's: in the failing
// build the instruction behind `ds_read_b96 v[18:20], v0 ; s_waitcnt lgkmcnt(0)` that first touches the LAST dword of the return (the
// centre's z, v20) is the v_mov_b32 that feeds the packed subtraction -- and both lanes caught were in 48-63, the last quarter a wide
// LDS return writes.  If, under LDS contention from a co-resident kernel, lgkmcnt could reach 0 a moment before the last quarter of the
// last dword has landed, those lanes would compute their distance against the PREVIOUS centre's z: too large a distance = a missed min,
// which is what the log showed.  (The plain build reads x, y, z in three dependent chains and touches z later.)
// The probe: every wave reads a float4 table entry by a broadcast ds_read_b96 at a changing, wave-uniform index, waits lgkmcnt(0) and
// copies the THIRD dword in the very next instruction; the copy is compared with the entry's known z, per lane quarter -- alone, and
// while a second kernel that lives on LDS traffic is co-resident on the same CUs (small blocks, 8 KB of LDS: both fit).
// NOT RUN YET: hipcc -O3 --offload-arch=gfx950 tools/tune/lds_return_probe.hip -o tools/tune/lds_return_probe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NPT = 2048;

__global__ __launch_bounds__(512) void probe(unsigned int *bad, int iters, int variant) {
  extern __shared__ float4 tab[];
  for (int k = threadIdx.x; k < NPT; k += 512) tab[k] = make_float4(k * 1.0f, k * 2.0f, k * 3.0f + 0.5f, k * 5.0f);
  __syncthreads();
  typedef __attribute__((address_space(3))) float4 lds_f4;
  const unsigned base = (unsigned)(uintptr_t)(lds_f4 *)&tab[0];
  unsigned far = (blockIdx.x * 977u + (threadIdx.x >> 6) * 131u) % NPT;  // wave-uniform
  unsigned wrong = 0;
  float keep = 0.f;
  for (int i = 0; i < iters; ++i) {
    const unsigned addr = base + 16u * far;
    float z, x;
    if (variant == 0) {  // b96, the failing build's read
      asm volatile("ds_read_b96 v[200:202], %2\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v202\n\tv_mov_b32 %1, v200"
                   : "=v"(z), "=v"(x) : "v"(addr) : "v200", "v201", "v202", "memory");
    } else {  // b128
      asm volatile("ds_read_b128 v[200:203], %2\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v202\n\tv_mov_b32 %1, v200"
                   : "=v"(z), "=v"(x) : "v"(addr) : "v200", "v201", "v202", "v203", "memory");
    }
    wrong += (z != far * 3.0f + 0.5f) + (x != far * 1.0f);
    keep += z;
    far = (far * 1103515245u + 12345u + (unsigned)i) % NPT;  // the next entry is a different one almost always
  }
  if (wrong) atomicAdd(&bad[(threadIdx.x & 63) >> 4], wrong);
  if (keep == -1.f) bad[7] = 1;  // keeps the loads alive
}

__global__ __launch_bounds__(256) void lds_noise(float *out, int iters) {
  __shared__ float s[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) s[i] = (float)i;
  __syncthreads();
  float a = 0.f;
  for (int i = 0; i < iters; ++i) {
    a += s[(threadIdx.x * 33 + i * 7) & 2047];  // conflicting reads
    s[(threadIdx.x * 32 + i * 13) & 2047] = a;  // and 32-way conflicting writes
  }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}

int main() {
  unsigned int *bad; float *scratch;
  (void)hipMalloc(&bad, 32); (void)hipMalloc(&scratch, 8192 * 256 * 4);
  hipStream_t s0, s1;
  (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1);
  for (int variant = 0; variant < 2; ++variant)
    for (int beside = 0; beside < 2; ++beside) {
      (void)hipMemset(bad, 0, 32);
      for (int rep = 0; rep < 10; ++rep) {
        if (beside) lds_noise<<<4096, 256, 0, s1>>>(scratch, 20000);
        probe<<<256, 512, NPT * sizeof(float4), s0>>>(bad, 100000, variant);  // one block per CU, 32 KB of LDS: the noise blocks fit beside it
      }
      (void)hipDeviceSynchronize();
      unsigned int r[8];
      (void)hipMemcpy(r, bad, 32, hipMemcpyDeviceToHost);
      printf("%s, %s: stale third/first dwords in lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u (of %.3g reads per quarter)\n",
             variant ? "ds_read_b128" : "ds_read_b96 ", beside ? "beside an LDS-bound kernel on the same CUs" : "alone", r[0], r[1], r[2], r[3],
             10.0 * 256 * 8 * 100000 * 16);
    }
  return 0;
}
