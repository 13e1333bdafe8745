// The failing build's instruction SEQUENCE, in isolation (docs/kernels/round6.md section 1): if neither tools/tune/pk_f32_probe.hip (H1: the
// v_mov_b32 -> v_pk_add_f32 pair) nor tools/tune/lds_return_probe.hip (H2: the last dword of the wide LDS return) shows a fault alone, the
// next candidate is their combination exactly as the failing fps_lean had it:
//     ds_read_b96 v[18:20], vA ; s_waitcnt lgkmcnt(0)
//     v_pk_add_f32 (x of two points - c.x) ; v_pk_add_f32 (y - c.y)
//     v_mov_b32 v0, v20                      ; c.z into the LOW half of the pair v[0:1] whose HIGH half is a live running value
//     v_pk_add_f32 v[20:21], vZ[0:1], v[0:1] op_sel_hi:[1,0] neg  ; z - c.z, read in the next slot
//     ... squares, sums, v_min_u32 v1, v21, v1 ; v_min_u32 v2, v20, v2
// One wave per SIMD (256-thread blocks, one per CU, like fps_lean at 4 waves per cloud), a table of "centres" in LDS, a changing
// wave-uniform index, and every lane checks its two distances against the same arithmetic done with plain instructions on a SEPARATE read
// of the table (ds_read_b32 x 3, waited for one by one).  Mismatches per lane quarter, alone and beside an LDS-bound co-resident kernel.
// NOT RUN YET: hipcc -O3 --offload-arch=gfx950 tools/tune/fps_seq_probe.hip -o tools/tune/fps_seq_probe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NPT = 2048;

__global__ __launch_bounds__(256) void probe(unsigned int *bad, int iters) {
  extern __shared__ float4 tab[];
  for (int k = threadIdx.x; k < NPT; k += 256) {
    const float f = (float)k;
    tab[k] = make_float4(__sinf(f * 0.37f), __cosf(f * 0.11f), __sinf(f * 0.05f + 1.f), 0.f);
  }
  __syncthreads();
  typedef __attribute__((address_space(3))) float4 lds_f4;
  const unsigned base = (unsigned)(uintptr_t)(lds_f4 *)&tab[0];
  const float px0 = threadIdx.x * 0.003f - 0.4f, px1 = px0 + 0.11f, py0 = 0.2f - threadIdx.x * 0.002f, py1 = py0 * 0.5f;
  const float pz0 = 0.05f * (threadIdx.x & 15), pz1 = pz0 - 0.3f;
  unsigned far = (blockIdx.x * 977u + (threadIdx.x >> 6) * 131u) % NPT;  // wave-uniform
  unsigned run0 = 0x501502f9u, run1 = 0x501502f9u, chk0 = run0, chk1 = run1;  // 1e10 as bits
  unsigned wrong = 0;
  for (int i = 0; i < iters; ++i) {
    const unsigned addr = base + 16u * far;
    // the sequence under test: v[200:202] = the centre, v204 = run1 (the live value in the high half of the pair v[203:204]... see below)
    // registers: v[210:211] = (px0, px1), v[212:213] = (py0, py1), v[214:215] = (pz0, pz1); the pair that takes c.z is v[206:207] with v207 = run1
    asm volatile(
        "v_mov_b32 v210, %2\n\tv_mov_b32 v211, %3\n\tv_mov_b32 v212, %4\n\tv_mov_b32 v213, %5\n\tv_mov_b32 v214, %6\n\tv_mov_b32 v215, %7\n\t"
        "v_mov_b32 v207, %1\n\t"   // run1 lives in the HIGH half of the pair
        "ds_read_b96 v[200:202], %8\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_add_f32 v[216:217], v[210:211], v[200:201] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 v[218:219], v[212:213], v[200:201] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_mov_b32 v206, v202\n\t"
        "v_pk_add_f32 v[220:221], v[214:215], v[206:207] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 v[216:217], v[216:217], v[216:217]\n\t"
        "v_pk_mul_f32 v[218:219], v[218:219], v[218:219]\n\t"
        "v_pk_mul_f32 v[220:221], v[220:221], v[220:221]\n\t"
        "v_pk_add_f32 v[216:217], v[216:217], v[218:219]\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 v[220:221], v[220:221], v[216:217]\n\t"
        "s_nop 0\n\t"
        "v_min_u32 v207, v221, v207\n\t"
        "v_min_u32 %0, v220, %0\n\t"
        "v_mov_b32 %1, v207"
        : "+v"(run0), "+v"(run1)
        : "v"(px0), "v"(px1), "v"(py0), "v"(py1), "v"(pz0), "v"(pz1), "v"(addr)
        : "v200", "v201", "v202", "v206", "v207", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220",
          "v221", "memory");
    // the same arithmetic on plain instructions from three separate, individually awaited reads
    float cx, cy, cz;
    asm volatile("ds_read_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %1, %3 offset:4\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %2, %3 offset:8\n\t"
                 "s_waitcnt lgkmcnt(0)\n\ts_nop 4"
                 : "=&v"(cx), "=&v"(cy), "=&v"(cz) : "v"(addr) : "memory");
    float d0, d1;
    {
      const float ax = px0 - cx, ay = py0 - cy, az = pz0 - cz, bx = px1 - cx, by = py1 - cy, bz = pz1 - cz;
      float a2, b2;
      asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3" : "=&v"(a2), "=&v"(b2) : "v"(ax), "v"(bx));
      float ay2, by2, az2, bz2;
      asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3" : "=&v"(ay2), "=&v"(by2) : "v"(ay), "v"(by));
      asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3" : "=&v"(az2), "=&v"(bz2) : "v"(az), "v"(bz));
      asm volatile("v_add_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(d0), "=&v"(d1) : "v"(a2), "v"(ay2), "v"(b2), "v"(by2));
      asm volatile("v_add_f32 %0, %2, %0\n\tv_add_f32 %1, %3, %1" : "+v"(d0), "+v"(d1) : "v"(az2), "v"(bz2));
    }
    const unsigned u0 = __float_as_uint(d0), u1 = __float_as_uint(d1);
    chk0 = u0 < chk0 ? u0 : chk0;
    chk1 = u1 < chk1 ? u1 : chk1;
    wrong += (run0 != chk0) + (run1 != chk1);
    if ((i & 63) == 63) run0 = run1 = chk0 = chk1 = 0x501502f9u;  // start a new "cloud": a lost update shows for at most 64 steps
    far = (far * 1103515245u + 12345u + (unsigned)i) % NPT;
  }
  if (wrong) atomicAdd(&bad[(threadIdx.x & 63) >> 4], wrong);
}

__global__ __launch_bounds__(256) void lds_noise(float *out, int iters) {
  __shared__ float s[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) s[i] = (float)i;
  __syncthreads();
  float a = 0.f;
  for (int i = 0; i < iters; ++i) {
    a += s[(threadIdx.x * 33 + i * 7) & 2047];
    s[(threadIdx.x * 32 + i * 13) & 2047] = a;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}

int main() {
  unsigned int *bad; float *scratch;
  (void)hipMalloc(&bad, 32); (void)hipMalloc(&scratch, 8192 * 256 * 4);
  hipStream_t s0, s1;
  (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1);
  for (int beside = 0; beside < 2; ++beside) {
    (void)hipMemset(bad, 0, 32);
    for (int rep = 0; rep < 10; ++rep) {
      if (beside) lds_noise<<<4096, 256, 0, s1>>>(scratch, 20000);
      probe<<<256, 256, NPT * sizeof(float4), s0>>>(bad, 100000);
    }
    (void)hipDeviceSynchronize();
    unsigned int r[8];
    (void)hipMemcpy(r, bad, 32, hipMemcpyDeviceToHost);
    printf("the failing sequence, %s: steps with a wrong running distance in lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u (of %.3g lane-steps per quarter)\n",
           beside ? "beside an LDS-bound kernel on the same CUs" : "alone", r[0], r[1], r[2], r[3], 10.0 * 256 * 4 * 100000 * 16);
  }
  return 0;
}
