#!/usr/bin/env python3
"""hitadv_group_add_relu_linear at cfg4's first level, a few launches: the target of the --pmc passes behind cfg4's roofline.traffic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.roofline_group_add_relu_linear(torch.device('cuda', 0), 64, 2048, 512, 32, 64, "PointNet++ sa1")
print(r)
