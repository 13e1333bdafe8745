cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
for o in 0 1 2; do for c in cfg4 cfg5; do HITADV_GAR_XCD=$o python - $c $o <<'PY'
import sys, os, torch
sys.path.insert(0, '.')
import bench
c = sys.argv[1]
dev = torch.device('cuda', 0)
r = bench.roofline_group_add_relu(dev, 64, 2048, 512, 32, 64, "PointNet++ sa1") if c == 'cfg4' else bench.roofline_group_add_relu(dev, 32, 512, 256, 32, 256, "PCT gather_local_1")
print("xcd_order", sys.argv[2], c, r['us_per_launch'], 'us', r['frac'])
PY
done; done
