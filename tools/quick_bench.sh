#!/bin/bash
# One headline run (cfg2, driver's flags, no cfg3-5 children, no CPU baseline) and the engine's kernel tests: the A/B loop of a
# kernel change.  usage: gpurun -- bash tools/quick_bench.sh <tag> [pytest -k expression]
tag=${1:-x}
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attack.py -q -m gpu -x ${2:+-k "$2"} 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/r04/qb_$tag.json 2> gpurun_out/r04/qb_$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04/qb_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["value"],2), "clouds/s", d["end_to_end"]["us_per_iteration"], "us/iter frac", d["end_to_end"]["frac"], "single", round(d["single_attack"]["value"],2))
PY
