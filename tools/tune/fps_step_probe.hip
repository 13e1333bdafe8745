// What each part of an FPS step's dependent chain costs (tools/tune/fps_lean_diag.hpp, the instrumented copy of sampling.hip::fps_lean, PROBE 1..5): one workgroup per cloud, the step is
// a serial chain on one CU, so the parts add.  hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950
// -I include -I hit_adv_amd/csrc tools/tune/fps_step_probe.hip -o tools/tune/fps_step_probe
#include "../../hit_adv_amd/csrc/sampling.hip"
#include "fps_lean_diag.hpp"
#include <cstdio>
#include <vector>
template <int PT, bool PCT, int NW, int PROBE>
static void run(const float *x, const int64_t *start, int64_t *idx, int B, int N, int m) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0, 0);
    hitadv::fps_lean_diag<PT, PCT, NW, int64_t, PROBE><<<B, 64 * NW, (size_t)N * sizeof(float4), 0>>>(x, start, N, m, idx);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("  N %5d PT %2d waves %d %s probe %d : %7.1f us  %.3f us/step\n", N, PT, NW, PCT ? "pct" : "sq ", PROBE, best * 1e3f, best * 1e3f / m);
}
template <int PT, bool PCT, int NW>
static void stamps(const float *x, const int64_t *start, int64_t *idx, int B, int N, int m) {
  hitadv::fps_lean_diag<PT, PCT, NW, int64_t, 5><<<B, 64 * NW, (size_t)N * sizeof(float4), 0>>>(x, start, N, m, idx);
  int64_t h[5];
  (void)hipMemcpy(h, idx, sizeof(h), hipMemcpyDeviceToHost);
  printf("  N %5d PT %2d waves %d %s shader cycles per step: centre read %.0f | distances + lane max %.0f | wave max + holder %.0f | slot write + barrier %.0f | slot read + merge %.0f\n",
         N, PT, NW, PCT ? "pct" : "sq ", (double)h[0] / m, (double)h[1] / m, (double)h[2] / m, (double)h[3] / m, (double)h[4] / m);
}
template <int PT, bool PCT, int NW>
static void all(const float *x, const int64_t *start, int64_t *idx, int B, int N, int m) {
  run<PT, PCT, NW, 0>(x, start, idx, B, N, m);
  run<PT, PCT, NW, 1>(x, start, idx, B, N, m);
  run<PT, PCT, NW, 2>(x, start, idx, B, N, m);
  run<PT, PCT, NW, 3>(x, start, idx, B, N, m);
  run<PT, PCT, NW, 4>(x, start, idx, B, N, m);
  stamps<PT, PCT, NW>(x, start, idx, B, N, m);
}
int main() {
  const int B = 32, NMAX = 4096, m = 512;
  std::vector<float> h((size_t)B * NMAX * 3);
  uint32_t s = 12345u;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 8388608.0f - 1.0f; }
  float *x; int64_t *start, *idx;
  (void)hipMalloc(&x, h.size() * 4); (void)hipMalloc(&start, B * 8); (void)hipMalloc(&idx, (size_t)B * m * 8);
  (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemset(start, 0, B * 8);
  all<2, false, 4>(x, start, idx, B, 512, m);
  all<4, false, 4>(x, start, idx, B, 1024, m);
  all<2, false, 8>(x, start, idx, B, 1024, m);
  all<8, false, 4>(x, start, idx, B, 2048, m);
  all<4, false, 8>(x, start, idx, B, 2048, m);
  all<2, false, 16>(x, start, idx, B, 2048, m);
  all<4, true, 4>(x, start, idx, B, 1024, m);
  all<2, true, 8>(x, start, idx, B, 1024, m);
  return 0;
}
