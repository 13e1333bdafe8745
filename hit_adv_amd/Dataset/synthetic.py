"""Synthetic stand-in for the ModelNet40 test split when no dataset is on the box: clouds with the
statistics ``pc_normalize`` leaves behind (centred, max-norm 1; Dataset/ModelNet.py:12-17 of the reference), unit normals
and a label in [0, 40).  Every cloud is a pure function of its id, so any rank of any run draws the same cloud."""
import torch
from torch.utils.data import Dataset


class SyntheticClouds(Dataset):
    """``kind='gaussian'``: iid normal points (the bench workload).  ``kind='sphere'``: points on the unit sphere with 1 %
    radial noise and the outward direction as normal -- a surface-like cloud, whose kNN statistics are closer to a scan."""

    def __init__(self, count, npoints=1024, kind='gaussian', first=0, num_class=40):
        if kind not in ('gaussian', 'sphere'):
            raise ValueError("kind must be 'gaussian' or 'sphere', got %r" % (kind,))
        self.count, self.npoints, self.kind, self.first, self.num_class = count, npoints, kind, first, num_class

    def __len__(self):
        return self.count

    def __getitem__(self, index):
        if not 0 <= index < self.count:
            raise IndexError(index)
        g = torch.Generator('cpu').manual_seed(1234 + self.first + index)
        xyz = torch.randn(self.npoints, 3, generator=g)
        if self.kind == 'gaussian':
            xyz = xyz - xyz.mean(0, keepdim=True)
            xyz = xyz / xyz.norm(dim=1).max()
            normal = torch.nn.functional.normalize(torch.randn(self.npoints, 3, generator=g), dim=1)
        else:
            normal = torch.nn.functional.normalize(xyz, dim=1)
            xyz = normal * (1. + 0.01 * torch.randn(self.npoints, 1, generator=g))
            xyz = xyz - xyz.mean(0, keepdim=True)
            xyz = xyz / xyz.norm(dim=1).max()
        label = torch.randint(0, self.num_class, (1,), generator=g)
        return torch.cat([xyz, normal], 1), label[0]


def synth_cloud(cloud_id, n):
    """One bench / fixture cloud: ``(xyz|normal [n,6], label [1])``, a pure function of its id."""
    g = torch.Generator('cpu').manual_seed(1234 + cloud_id)
    xyz = torch.randn(n, 3, generator=g)
    xyz = xyz - xyz.mean(0, keepdim=True)
    xyz = xyz / xyz.norm(dim=1).max()
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    label = torch.randint(0, 40, (1,), generator=g)
    return torch.cat([xyz, nrm], 1), label


def synth_batch(b, n, first=0):
    """``b`` consecutive clouds starting at id ``first``: ``(data [b,n,6], label [b])``."""
    cl = [synth_cloud(first + i, n) for i in range(b)]
    return torch.stack([c[0] for c in cl]), torch.cat([c[1] for c in cl])


class ToyVictim(torch.nn.Module):
    """The < 1K-parameter victim (shared 3->16 layer, max over points, 16->40) that smoke runs and fixtures carry."""

    def __init__(self, classes=40, width=16):
        super().__init__()
        self.conv = torch.nn.Conv1d(3, width, 1)
        self.fc = torch.nn.Linear(width, classes)

    def forward(self, x):
        h = torch.relu(self.conv(x))
        return self.fc(torch.max(h, 2)[0])


def shake_bn(model, seed=1, mean_std=0.05, var_spread=0.2):
    """Running statistics of every BatchNorm layer away from their initial 0 / 1, from one CPU generator in module order: a
    random-init victim in eval mode whose normalisation layers are not identities (what a trained checkpoint looks like to
    the attack).  A pure function of (architecture, seed, mean_std, var_spread): fixtures store the three numbers, not the
    statistics.  Call before moving the model to a device."""
    g = torch.Generator('cpu').manual_seed(int(seed))
    bn = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, bn):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * mean_std)
                mod.running_var.copy_(1. - var_spread + 2. * var_spread * torch.rand(mod.running_var.shape, generator=g))
    return model


def sphere_batch(b, n, first=0):
    """``b`` surface-like clouds (``SyntheticClouds(kind='sphere')``) starting at id ``first``: ``(data [b,n,6], label [b])``."""
    ds = SyntheticClouds(b, n, kind='sphere', first=first)
    rows = [ds[i] for i in range(b)]
    return torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows])


def sharpen(model, gain):
    """Every convolution / linear WEIGHT (not the biases) times ``gain``.  A default-initialised deep victim in eval mode
    computes logits that are its last layer's bias plus a feature term a hundred times smaller: every cloud lands in the
    class of the largest bias and no bounded deformation moves it (0 / 256 successes on PointNet++, 0 / 96 on PCT in
    round 4's bench lines).  A gain > 1 compounds through the layers until the feature term decides the class, which is
    what a trained victim looks like to an attack.  The toy victims of fixtures g5 / g7 are made the same way
    (``conv.weight.mul_(3.0)``)."""
    kinds = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Linear)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, kinds):
                mod.weight.mul_(gain)
    return model
