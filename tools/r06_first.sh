#!/bin/bash
# round 6, first lease (refused so far: GPU use for the repository was closed from outside the build -- docs/kernels/round6.md section 1): the probes round 5 wrote and never ran + this round's
# second hypothesis, then the driver's three checks on HEAD's build.  Outputs: gpurun_out/r06/ (copy what is to be judged to profiles/r06_*).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/tune/pk_f32_probe.hip -o /tmp/pk_f32_probe 2>&1 | tail -3
timeout 300 /tmp/pk_f32_probe > gpurun_out/r06/pk_f32_probe.txt 2>&1; echo "pk probe rc=$?" >> gpurun_out/r06/pk_f32_probe.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/tune/lds_return_probe.hip -o /tmp/lds_return_probe 2>&1 | tail -3
timeout 300 /tmp/lds_return_probe > gpurun_out/r06/lds_return_probe.txt 2>&1; echo "lds probe rc=$?" >> gpurun_out/r06/lds_return_probe.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/tune/fps_seq_probe.hip -o /tmp/fps_seq_probe 2>&1 | tail -3
timeout 300 /tmp/fps_seq_probe > gpurun_out/r06/fps_seq_probe.txt 2>&1; echo "seq probe rc=$?" >> gpurun_out/r06/fps_seq_probe.txt

timeout 1500 bash tools/fps_packed_repro.sh > gpurun_out/r06/fps_packed_repro.txt 2>&1; echo "repro rc=$?" >> gpurun_out/r06/fps_packed_repro.txt

timeout 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r06/pytest_gpu_first.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r06/pytest_gpu_first.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_first.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r06/smoke_first.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_first.txt 2>&1; echo "bench rc=$?" >> gpurun_out/r06/bench_first.txt
# the A/B that decides about the deferred-search V1 (docs/kernels/round6.md section 4): its bitwise test first, then the headline either way, same box
timeout 600 python -m pytest tests/test_z_r06_v1_defer.py -q > gpurun_out/r06/v1_defer_test.txt 2>&1; echo "defer test rc=$?" >> gpurun_out/r06/v1_defer_test.txt
for d in 0 1 0 1; do
  HITADV_V1_DEFER=$d timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-single --no-f32 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('HITADV_V1_DEFER=$d', d['value'], d['ms_per_step'], d.get('attack_success'))" >> gpurun_out/r06/v1_defer_ab.txt 2>&1
done
tail -3 gpurun_out/r06/*.txt
