"""PointNet victim (stays on PyTorch-ROCm: rocBLAS / MIOpen run its forward and backward).

Parameter and buffer names are those of the reference's model/feature_models.py
(PointNetFeatureModel :71-98, PointNetEncoder :101-147, STN3d :150-187, STNkd :190-230), so
``load_state_dict(torch.load('PN_NT.checkpoint')['model_state_dict'])`` (eval.py:79,123) works
unchanged: 111 state_dict entries, 3,471,473 parameters (tests/golden/g8_state_dicts.json).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _TNet(nn.Module):
    """Spatial transformer: three shared 1x1 convs -> max over points -> three FC -> (I + A)."""

    def __init__(self, in_ch, out_dim):
        super().__init__()
        self.out_dim = out_dim
        self.conv1 = nn.Conv1d(in_ch, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, out_dim * out_dim)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)

    def forward(self, x):
        h = F.relu(self.bn1(self.conv1(x)))
        h = F.relu(self.bn2(self.conv2(h)))
        h = F.relu(self.bn3(self.conv3(h)))
        h = h.max(dim=2)[0]
        h = F.relu(self.bn4(self.fc1(h)))
        h = F.relu(self.bn5(self.fc2(h)))
        h = self.fc3(h)
        eye = torch.eye(self.out_dim, device=h.device, dtype=h.dtype).reshape(1, -1)
        return (h + eye).view(-1, self.out_dim, self.out_dim)


class STN3d(_TNet):
    def __init__(self, channel):
        super().__init__(channel, 3)


class STNkd(_TNet):
    def __init__(self, k=64):
        super().__init__(k, k)
        self.k = k


class PointNetEncoder(nn.Module):
    def __init__(self, global_feat=True, feature_transform=False, channel=3):
        super().__init__()
        self.stn = STN3d(channel)
        self.conv1 = nn.Conv1d(channel, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.global_feat = global_feat
        self.feature_transform = feature_transform
        if feature_transform:
            self.fstn = STNkd(k=64)

    def forward(self, x):
        B, D, N = x.shape
        trans = self.stn(x)
        pts = x.transpose(2, 1)
        if D > 3:
            extra = pts[:, :, 3:]
            pts = pts[:, :, :3]
        pts = torch.bmm(pts, trans)
        if D > 3:
            pts = torch.cat([pts, extra], dim=2)
        h = F.relu(self.bn1(self.conv1(pts.transpose(2, 1))))
        trans_feat = None
        if self.feature_transform:
            trans_feat = self.fstn(h)
            h = torch.bmm(h.transpose(2, 1), trans_feat).transpose(2, 1)
        point_feat = h
        h = F.relu(self.bn2(self.conv2(h)))
        h = self.bn3(self.conv3(h))
        g = h.max(dim=2)[0].view(-1, 1024)
        if self.global_feat:
            return g, trans, trans_feat
        return torch.cat([g.view(-1, 1024, 1).repeat(1, 1, N), point_feat], 1), trans, trans_feat


class PointNetFeatureModel(nn.Module):
    """The classifier eval.py:109 builds: ``forward(x[B,3,N]) -> (logits[B,k], trans_feat)``."""

    def __init__(self, k=40, normal_channel=True):
        super().__init__()
        self.feat = PointNetEncoder(global_feat=True, feature_transform=True,
                                    channel=6 if normal_channel else 3)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k)
        self.dropout = nn.Dropout(p=0.4)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()
        self.eval()

    def forward(self, x):
        g, _, trans_feat = self.feat(x)
        h = F.relu(self.bn1(self.fc1(g)))
        h = F.relu(self.bn2(self.dropout(self.fc2(h))))
        return self.fc3(h), trans_feat
