"""Shared test helpers: fixture loading, synthetic inputs, the toy victim."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def T(a):
    t = torch.from_numpy(np.asarray(a))
    return t


def synth_cloud(cloud_id, n):
    """Synthetic input of BASELINE.md section 3 (same rule as tests/golden/make_golden.py)."""
    g = torch.Generator('cpu').manual_seed(1234 + cloud_id)
    xyz = torch.randn(n, 3, generator=g)
    xyz = xyz - xyz.mean(0, keepdim=True)
    xyz = xyz / xyz.norm(dim=1).max()
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    label = torch.randint(0, 40, (1,), generator=g)
    return torch.cat([xyz, nrm], 1), label


def synth_batch(b, n, first=0):
    cl = [synth_cloud(first + i, n) for i in range(b)]
    return torch.stack([c[0] for c in cl]), torch.cat([c[1] for c in cl])


class ToyVictim(torch.nn.Module):
    """The < 1K-parameter victim whose weights travel inside fixtures g5/g5b/g7."""

    def __init__(self, classes=40, width=16):
        super().__init__()
        self.conv = torch.nn.Conv1d(3, width, 1)
        self.fc = torch.nn.Linear(width, classes)

    def forward(self, x):
        h = torch.relu(self.conv(x))
        return self.fc(torch.max(h, 2)[0])


def toy_from_fixture(fx):
    m = ToyVictim()
    m.load_state_dict({k[2:]: T(v) for k, v in fx.items() if k.startswith('w_')})
    return m.eval()


def hp_from_fixture(fx):
    hp = {}
    for k, v in fx.items():
        if k.startswith('hp_'):
            v = v.item()
            hp[k[3:]] = int(v) if float(v).is_integer() and k[3:] in (
                'central_num', 'total_central_num', 'binary_step', 'num_iter', 'curv_loss_knn') else float(v)
    return hp
