// libfps_diag.so: the instrumented fps_lean (tools/tune/fps_lean_diag.hpp) behind a C entry point, so that tools/fps_check.py can run
// it beside the product library's kernels without the product library carrying any of it.  Built by tools/fps_packed_repro.sh:
//   hipcc -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -I include -I hit_adv_amd/csrc
//         -DHITADV_FPS_DIAG=<0|2|4|5|9> tools/tune/fps_diag_lib.hip -o tools/build/libfps_diag_<n>.so
// MODE 0 only (HiT_ADV.py:489-510 / pointnet2_utils.py:63-84 semantics), the sizes fps_lean takes (256 < N <= 4080).
#include "../../hit_adv_amd/csrc/sampling.hip"
#include "fps_lean_diag.hpp"

static unsigned long long *g_log = nullptr;

extern "C" int fpsdiag_log(unsigned long long *dev_log) {  // HITADV_FPS_DIAG == 9: where the next launch logs its waves' keys
  g_log = dev_log;
  return 0;
}

extern "C" int fpsdiag_counters(unsigned int *host8) {
  return (int)hipMemcpyFromSymbol(host8, HIP_SYMBOL(hitadv::g_fps_dbg), sizeof(unsigned int) * 8);
}

extern "C" int fpsdiag_build(void) { return HITADV_FPS_DIAG; }

extern "C" int fpsdiag_fps_from_start(const float *xyz, const int64_t *start, int B, int N, int m, int64_t *idx, void *stream) {
  if (!xyz || !start || !idx || B <= 0 || N <= 256 || N > 4080 || m <= 0) return HITADV_E_ARG;
  const size_t shm = (size_t)N * sizeof(float4);
  hipStream_t s = (hipStream_t)stream;
#define DIAG_CASE(PT, NW)                                                                                  \
  if (N <= 64 * NW * PT) {                                                                                 \
    hitadv::fps_lean_diag<PT, false, NW, int64_t><<<B, 64 * NW, shm, s>>>(xyz, start, N, m, idx, g_log);    \
    return (int)hipGetLastError();                                                                         \
  }
  if (N > 512) {  // the launcher's shape: eight waves above 512 points
    DIAG_CASE(2, 8)
    DIAG_CASE(4, 8)
    DIAG_CASE(8, 8)
  } else {
    DIAG_CASE(2, 4)
  }
#undef DIAG_CASE
  return HITADV_E_ARG;
}
