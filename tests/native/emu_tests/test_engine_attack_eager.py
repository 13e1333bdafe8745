"""EMULATOR-ONLY (run by tools/run_emulated_suite.sh under tests/native/emu_plugin.py; ~7 min): tests/test_gpu_attack.py::
test_hit_adv_pointnet_engine_follows_the_cpu_oracle with the hipGraph switched off -- HiT-ADV with the PointNet HIP ENGINE (default matrix
mode fp16x2: V1, the streaming V2 with the deformation inside the first kernel, V3, the FC chains, the iteration head, the Adam tail; every
matrix instruction emulated) for 2 x 8 iterations at B = 3, N = 256 against the CPU oracle driving the plain nn.Module: the last iterate
and the returned clouds within the GPU test's own tolerance (the emulator achieves 2.4e-7, the figure that test's comment records for
MI355X), the same success count."""
import contextlib
import copy
import io
import os

import numpy as np
import pytest
import torch

from helpers import synth_batch
from oracle import hitadv_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("HITADV_EMULATE") != "1", reason="emulator-only: on a GPU the suite's own tests cover this")]


def test_hit_adv_with_the_pointnet_engine_eager_follows_the_cpu_oracle():
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    torch.manual_seed(3)
    cpu_model = PointNetFeatureModel(40, normal_channel=False).eval()
    with torch.no_grad():
        for mod in cpu_model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.05)
                mod.running_var.uniform_(0.8, 1.2)
    gpu_model = copy.deepcopy(cpu_model)
    data, _ = synth_batch(3, 256, first=3000)
    with torch.no_grad():
        label = cpu_model(data[:, :, :3].transpose(1, 2).contiguous())[0].argmax(1)
    hp = dict(binary_step=2, num_iter=8, cd_weight=1e-4, ker_weight=1., hide_weight=1., curv_loss_knn=16, central_num=32,
              total_central_num=64, max_sigm=1.2, min_sigm=0.1, budget=0.55)
    att = HiT_ADV(gpu_model, UntargetedLogitsAdvLoss(30.), verbose=False, use_graph=False, **hp)
    torch.manual_seed(12)
    best, succ = att.attack(data, label)
    assert att._view is not None and att._view.hip_engine and att._view.matrix_mode == 'fp16x2'
    ws = next(iter(att._ws.values()))
    trace = []
    oracle = O.HiTADVOracle(cpu_model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), **hp)
    torch.manual_seed(12)
    with contextlib.redirect_stdout(io.StringIO()):
        obest, osucc = oracle.attack(data, label, trace=trace)
    assert len(trace) == 16
    np.testing.assert_allclose(ws.adv.numpy(), trace[-1]['adv'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(best, obest, rtol=1e-4, atol=1e-5)
    assert int(succ) == int(osucc)
