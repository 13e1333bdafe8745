#!/usr/bin/env python3
"""Only the two matrix-core kernels that dominate cfg2 / cfg3, a few launches each -- meant for rocprofv3 --pmc passes
(tools/r03_measure.sh pmc_mfma): V1 (hitadv_linear_max_fwd_f16x2_packed, 128 -> 1024 + max over 1024 points, at B = 32 and at
the stacked B = 128) and G16 (hitadv_linear_lrelu_pool_fwd, 32 x 1024 points, 512 -> 1024)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import ops  # noqa: E402
from hit_adv_amd.model.pointnet import PointNetFeatureModel  # noqa: E402

torch.manual_seed(0)
view = PointNetFeatureModel(40, normal_channel=False).cuda().eval().attack_view()
g = torch.Generator().manual_seed(0)
N = 1024
for B in (32, 128):
    x = (torch.randn(B, 3, N, generator=g) * 0.4).cuda()
    a1, a2 = torch.empty(B * N, 64, device='cuda'), torch.empty(B * N, 128, device='cuda')
    ops.pointnet_rowmlp_fwd(0, B, N, view.s2_w, view.s2_b, a2, x=x, W0=view.s1_w, b0=view.s1_b, o0=a1, mode=2, range_flag=view.range_flag)
    for _ in range(5):
        ops.linear_max_fwd_f16x2(a2, view.pieces('s3', 2), B, N, bias=view.s3_b, relu=True, blocks=(128 if B > 32 else 0),
                                 range_flag=view.range_flag, packed=True)
Bc, Cin, C = 32, 512, 1024
xa = torch.randn(Bc * N, Cin, generator=g).cuda()
W = (torch.randn(C, Cin, generator=g) / Cin ** 0.5).cuda()
bias = torch.randn(C, generator=g).cuda()
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
Wp, Wtp = ops.split_rows_f16x2(W, flag), ops.split_rows_f16x2(W.t().contiguous(), flag)
for _ in range(5):
    ops.linear_lrelu_pool(xa, Wp, Wtp, bias, Bc, N, 0.2, flag)
torch.cuda.synchronize()
