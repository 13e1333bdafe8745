#!/bin/bash
# gpurun -- bash tools/fps_packed_repro.sh : docs/kernels/round5.md section 8 again, WITHOUT touching the product library.  Builds
# tools/tune/fps_diag_lib.hip (the instrumented copy of fps_lean) twice -- distances on packed f32 instructions (-DHITADV_FPS_DIAG=2: the
# same eight packed instructions per pair of points as the build that failed) and on plain ones (0) -- and runs tools/fps_check_modes.py
# with each as the "fps_lean" of the cross-check: with two attacks in flight the packed build's first launch differed from the
# 64-bit-key kernel for 1-30 % of the clouds, the plain build's never.  NOT RUN YET: round 5's GPU access closed before it was
# written, round 6 had none (docs/kernels/round6.md section 1).
cd "$GRAFT_REPO_ROOT"
mkdir -p tools/build
FLAGS="-O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Iinclude -Ihit_adv_amd/csrc -Wno-unused-result"
for d in 2 0; do
  /opt/rocm/bin/hipcc $FLAGS -DHITADV_FPS_DIAG=$d tools/tune/fps_diag_lib.hip -o tools/build/libfps_diag_$d.so 2>&1 | grep -E "error" -A5
  HITADV_FPS_DIAG_LIB="$GRAFT_REPO_ROOT/tools/build/libfps_diag_$d.so" timeout 600 python tools/fps_check_modes.py 2>&1 | grep "in flight" | sed "s/^/HITADV_FPS_DIAG=$d /" | cut -c1-200
done
