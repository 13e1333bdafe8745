#!/bin/bash
# gpurun -- bash tools/fps_packed_repro.sh : docs/kernels/round5.md section 8 again.  Builds the library with fps_lean's distances on packed
# f32 instructions (-DHITADV_FPS_DIAG=2) and with plain ones, and runs tools/fps_check_modes.py on both: with two attacks in flight the
# packed build's first launch differs from the 64-bit-key kernel for 1-30 % of the clouds, the plain build's never.
# (The numbers in the notes come from the build this flag re-creates -- the same eight packed instructions per pair of points in the ISA --;
# the flag itself was added after the round's GPU access had closed and has not been run.)
cd "$GRAFT_REPO_ROOT/hit_adv_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
for d in 2 0; do
  touch sampling.hip; make -s -j8 CXXFLAGS="-DHITADV_FPS_DIAG=$d $FLAGS" 2>&1 | grep -E "error" -A5
  (cd "$GRAFT_REPO_ROOT" && timeout 600 python tools/fps_check_modes.py 2>&1 | grep "in flight" | sed "s/^/HITADV_FPS_DIAG=$d /" | cut -c1-140)
done
