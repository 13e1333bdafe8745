"""PCT victim (cfg5 of BASELINE.json).  Parameter names follow the reference's model/pct_cls.py (Pct :27-75,
Local_op :6-24, Point_Transformer_Last :77-109, SA_Layer :111-139): 113 state_dict entries
(tests/golden/g8_state_dicts.json).  Attention / MLPs stay PyTorch-ROCm (every 1x1 convolution as a GEMM with its
BatchNorm folded in eval mode, see _pointwise.py); the neighbourhood construction of
``sample_and_group`` (model/pct_utils.py:111-140) runs in HIP:

* ``fps`` (util/other_utils.py:254-272): random first index from the CPU generator (:264) or from an attack's
  pre-drawn feed (_sampling.py), then ``hitadv_fps_pct``: running distances = sqrt of the clamped Gram-form distance
  in torch's own fp32 arithmetic (``get_dists``, :237-251), so the table is the reference's bit for bit (fixture g12);
* ``knn_point`` (pct_utils.py:98-109, ``topk(..., sorted=False)``): ``hitadv_knn_points`` on the reference's Gram-form
  ``square_distance`` values (sorted; the consumer max-pools over the neighbours, so order is irrelevant -- the SET is
  the reference's).
``ops.victim_reference_arithmetic(False)`` selects direct-form distances for both.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import _sampling
from ._pointwise import (conv1x1, fast_pm, linear_lrelu_maxpool_pm, linear_pm, linear_relu_max_pm, linear_relu_pm,
                         split_first_layer)
from .pointnet2 import index_points


def fps(xyz, M):
    B, N, _ = xyz.shape
    return ops.fps_pct(xyz, M, _sampling.next_start(B, N, xyz.device))


def knn_point(nsample, xyz, new_xyz):
    form = ops.FORM_SQUARE_DISTANCE if ops.victim_reference_arithmetic.get() else ops.FORM_DIRECT
    return ops.KnnPoints.apply(new_xyz.detach(), xyz.detach(), int(nsample), form)[1]


def sample_and_group(npoint, radius, nsample, xyz, points):
    """xyz [B,N,3], points [B,N,D] -> (new_xyz [B,S,3], features [B,S,nsample,2D])."""
    B, N, C = xyz.shape
    xyz = xyz.contiguous()
    fps_idx = fps(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    centre = index_points(points, fps_idx)
    idx = knn_point(nsample, xyz, new_xyz)
    rel = index_points(points, idx) - centre.view(B, npoint, 1, -1)
    return new_xyz, torch.cat([rel, centre.view(B, npoint, 1, -1).expand(-1, -1, nsample, -1)], dim=-1)


class Local_op(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv1d(in_channels, out_channels, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(out_channels, out_channels, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm1d(out_channels)
        self.bn2 = nn.BatchNorm1d(out_channels)

    def from_points(self, xyz, points, npoint, nsample, tables=None):
        """sample_and_group + forward without the grouped tensor (eval mode on the GPU): xyz [B,N,3], points [B,N,D] ->
        (new_xyz [B,S,3], features [B,S,C] points-major).  The first convolution is split over the neighbour and the
        centre, W [x_j - c_i ; c_i] = Wa x_j + (Wb - Wa) c_i: one GEMM over the N points, one over the S centres and
        ``hitadv_group_add_relu`` replace the [B,S,nsample,2D] gather / subtract / concat and a GEMM over S*nsample rows.
        Draws the FPS start exactly where ``sample_and_group`` does.  ``tables`` = (fps_idx, new_xyz, idx, ready) when the
        caller has the sampling / grouping tables computed ahead on another stream (``Pct._tables_ahead``): the per-point
        product below does not need them and runs meanwhile."""
        xyz = xyz.contiguous()
        D = points.shape[-1]
        W, t = split_first_layer(self.conv1, self.bn1, D)
        if tables is None:
            fps_idx = fps(xyz, npoint)
            new_xyz = index_points(xyz, fps_idx)
            idx = knn_point(nsample, xyz, new_xyz)
            U = torch.matmul(points, W[:, :D].t())
        else:
            U = torch.matmul(points, W[:, :D].t())
            fps_idx, new_xyz, idx, ready = tables
            torch.cuda.current_stream().wait_event(ready)
        V = torch.addmm(t, index_points(points, fps_idx).reshape(-1, D), (W[:, D:] - W[:, :D]).t()).view(-1, npoint, W.shape[0])
        return new_xyz, linear_relu_max_pm(self.conv2, self.bn2, ops.group_add_relu(U, V, idx))

    def fast(self, points, nsample):
        return (fast_pm(self.conv1, self.bn1, points) and self.conv1.in_channels == 2 * points.shape[-1] and
                ops.group_add_relu_supported(self.conv1.out_channels, nsample))

    def forward(self, x):
        b, n, s, d = x.shape
        if fast_pm(self.conv1, self.bn1, x):  # x is already points-major: two GEMMs and a max over the neighbours
            h = linear_relu_pm(self.conv2, self.bn2, linear_relu_pm(self.conv1, self.bn1, x))
            return h.max(dim=2)[0].permute(0, 2, 1)
        h = x.permute(0, 1, 3, 2).reshape(-1, d, s)
        h = F.relu(self.bn1(self.conv1(h)))
        h = F.relu(self.bn2(self.conv2(h)))
        return F.adaptive_max_pool1d(h, 1).view(b, n, -1).permute(0, 2, 1)


class SA_Layer(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.q_conv = nn.Conv1d(channels, channels // 4, 1, bias=False)
        self.k_conv = nn.Conv1d(channels, channels // 4, 1, bias=False)
        self.q_conv.weight = self.k_conv.weight  # shared, as in the reference (:116)
        self.q_conv.bias = self.k_conv.bias
        self.v_conv = nn.Conv1d(channels, channels, 1)
        self.trans_conv = nn.Conv1d(channels, channels, 1)
        self.after_norm = nn.BatchNorm1d(channels)
        self.act = nn.ReLU()
        self.softmax = nn.Softmax(dim=-1)

    # one autograd node per layer where the shapes allow (False, or HITADV_PCT_FUSED_SA=0: the op-by-op composition below)
    FUSED_BACKWARD = os.environ.get("HITADV_PCT_FUSED_SA", "1") != "0"

    def _fused_layer_ok(self, x):
        from . import _pointwise as P
        C = x.shape[2]
        return (self.FUSED_BACKWARD and not P.WEIGHT_GRADS and not self.training and self.v_conv.bias is not None
                and x.dtype == torch.float32 and ops.offset_attention_layer_supported(x.shape[1], C, self.q_conv.weight.shape[0]))

    def forward_pm(self, x):
        """The same layer on points-major x [B,N,C] (eval mode on the GPU): the 1x1 convolutions are GEMMs over B*N rows,
        the attention products take their transposes through the BLAS flags -- no permuted copies in either direction."""
        if x.is_cuda and x.dim() == 3 and self._fused_layer_ok(x):
            from . import _pointwise as P
            Wq, _ = P._folded(self.q_conv, None)
            Wv, bv = P._folded(self.v_conv, None)
            Wt, bt = P._folded(self.trans_conv, self.after_norm)
            # one autograd node with a hand-written backward pass (ops.OffsetAttentionLayer): the constants enter detached
            return ops.offset_attention_layer(x, Wq.detach(), Wv.detach(), bv.detach(), Wt.detach(), bt.detach())
        q = linear_pm(self.q_conv, None, x)  # q_conv and k_conv share their weight (:116): one product serves both
        # q q^T: the tiled batched kernel (csrc/bmm.hip); softmax + column renormalisation: csrc/attention.hip
        attention = ops.offset_attention_norm(ops.bmm(q, q, False, True))
        x_r = ops.bmm(attention, linear_pm(self.v_conv, None, x), True, False)  # attention^T v
        return x + linear_relu_pm(self.trans_conv, self.after_norm, x - x_r)

    def forward(self, x):
        attention = self.softmax(torch.bmm(conv1x1(self.q_conv, None, x).permute(0, 2, 1), conv1x1(self.k_conv, None, x)))
        attention = attention / (1e-9 + attention.sum(dim=1, keepdim=True))
        x_r = torch.bmm(conv1x1(self.v_conv, None, x), attention)
        return x + self.act(conv1x1(self.trans_conv, self.after_norm, x - x_r))


class Point_Transformer_Last(nn.Module):
    def __init__(self, args, channels=256):
        super().__init__()
        self.args = args
        self.conv1 = nn.Conv1d(channels, channels, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(channels, channels, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm1d(channels)
        self.bn2 = nn.BatchNorm1d(channels)
        self.sa1, self.sa2 = SA_Layer(channels), SA_Layer(channels)
        self.sa3, self.sa4 = SA_Layer(channels), SA_Layer(channels)

    def forward_pm(self, x):
        """x [B,N,C] points-major -> [B,N,4C]."""
        h = linear_relu_pm(self.conv2, self.bn2, linear_relu_pm(self.conv1, self.bn1, x))
        outs = []
        for sa in (self.sa1, self.sa2, self.sa3, self.sa4):
            h = sa.forward_pm(h)
            outs.append(h)
        return torch.cat(outs, dim=2)

    def forward(self, x):
        h = F.relu(conv1x1(self.conv2, self.bn2, F.relu(conv1x1(self.conv1, self.bn1, x))))
        outs = []
        for sa in (self.sa1, self.sa2, self.sa3, self.sa4):
            h = sa(h)
            outs.append(h)
        return torch.cat(outs, dim=1)


class Pct(nn.Module):
    def __init__(self, args, output_channels=40):
        super().__init__()
        self.args = args
        self.conv1 = nn.Conv1d(3, 64, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(64, 64, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(64)
        self.gather_local_0 = Local_op(in_channels=128, out_channels=128)
        self.gather_local_1 = Local_op(in_channels=256, out_channels=256)
        self.pt_last = Point_Transformer_Last(args)
        self.conv_fuse = nn.Sequential(nn.Conv1d(1280, 1024, kernel_size=1, bias=False), nn.BatchNorm1d(1024),
                                       nn.LeakyReLU(negative_slope=0.2))
        self.linear1 = nn.Linear(1024, 512, bias=False)
        self.bn6 = nn.BatchNorm1d(512)
        self.dp1 = nn.Dropout(p=args.dropout)
        self.linear2 = nn.Linear(512, 256)
        self.bn7 = nn.BatchNorm1d(256)
        self.dp2 = nn.Dropout(p=args.dropout)
        self.linear3 = nn.Linear(256, output_channels)

    def fps_start_plan(self, N):
        """One ``randint(0, high, (B,))`` per FPS call of a forward pass, in call order: 512 of the N points, then 256
        of those 512 (:62,65)."""
        return [N, 512]

    # the FPS / kNN chain of both Local_ops on a second stream.  OFF: measured on MI355X (tools/stream_overlap_probe.py), the
    # forked pass costs 7.8-8.1 ms as a captured graph against 3.63 ms in line -- the runtime executes a graph with parallel
    # branches segment by segment with cross-stream waits -- and 3.5 vs 2.3 ms per pass with three such graphs in flight.
    # Kept as a switch (the eager path does overlap: FPS 6.5 ms + a GEMM stream 16.2 ms run in 17.7 ms together).
    tables_ahead = False
    _side = {}

    def _tables_ahead(self, xyz):
        """Both Local_ops' sampling and grouping tables -- FPS 512 of N, kNN 32; FPS 256 of those 512, kNN 32 -- depend on the
        coordinates alone, and FPS is a serial chain of one workgroup per cloud (32 of 256 CUs busy for ~0.45 ms per pass at
        B = 32): they run on a second stream while the first per-point layers and the first Local_op's products use the
        rest of the chip.  Inside a captured iteration this becomes a fork / join of the graph.  Same calls in the same
        order as in line (the FPS starts are read from the feed in call order); nothing here carries gradient."""
        dev = xyz.device
        cur = torch.cuda.current_stream(dev)
        side = Pct._side.get(dev)
        if side is None:
            side = Pct._side[dev] = torch.cuda.Stream(dev)
        pts = xyz.detach()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            f0 = fps(pts, 512)
            c0 = index_points(pts, f0)
            i0 = knn_point(32, pts, c0)
            e0 = torch.cuda.Event()
            e0.record(side)
            f1 = fps(c0, 256)
            c1 = index_points(c0, f1)
            i1 = knn_point(32, c0, c1)
            e1 = torch.cuda.Event()
            e1.record(side)
        if not torch.cuda.is_current_stream_capturing():  # eager: the allocator must know the other stream uses them
            pts.record_stream(side)
            for t in (f0, c0, i0, f1, c1, i1):
                t.record_stream(cur)
        return (f0, c0, i0, e0), (f1, c1, i1, e1)

    def _forward_points_major(self, x):
        """Eval mode on the GPU: every tensor stays points-major [B,N,C], so each 1x1 convolution is one GEMM with the
        BatchNorm folded in and bias / ReLU in its epilogue, and nothing is permuted or copied between layers."""
        xyz = ops.points_major(x)
        t0, t1 = self._tables_ahead(xyz) if self.tables_ahead else (None, None)
        h = linear_relu_pm(self.conv2, self.bn2, linear_relu_pm(self.conv1, self.bn1, xyz))
        new_xyz, p0 = self.gather_local_0.from_points(xyz, h, 512, 32, tables=t0)
        new_xyz, p1 = self.gather_local_1.from_points(new_xyz, p0, 256, 32, tables=t1)
        # conv_fuse + BatchNorm + LeakyReLU + the max over the points: one kernel where the widths allow (_pointwise)
        g = linear_lrelu_maxpool_pm(self.conv_fuse[0], self.conv_fuse[1], torch.cat([self.pt_last.forward_pm(p1), p1], dim=2))
        g = self.dp1(F.leaky_relu(self.bn6(self.linear1(g)), negative_slope=0.2))
        g = self.dp2(F.leaky_relu(self.bn7(self.linear2(g)), negative_slope=0.2))
        return self.linear3(g)

    def forward(self, x):
        B = x.shape[0]
        if fast_pm(self.conv1, self.bn1, x) and not self.training:
            return self._forward_points_major(x)
        xyz = x.permute(0, 2, 1)
        h = F.relu(conv1x1(self.conv2, self.bn2, F.relu(conv1x1(self.conv1, self.bn1, x)))).permute(0, 2, 1)
        if self.gather_local_0.fast(h, 32):
            new_xyz, p0 = self.gather_local_0.from_points(xyz, h, 512, 32)
            new_xyz, p1 = self.gather_local_1.from_points(new_xyz, p0, 256, 32)
            f1 = p1.permute(0, 2, 1)
        else:
            new_xyz, grouped = sample_and_group(npoint=512, radius=0.15, nsample=32, xyz=xyz, points=h)
            f0 = self.gather_local_0(grouped)
            new_xyz, grouped = sample_and_group(npoint=256, radius=0.2, nsample=32, xyz=new_xyz,
                                                points=f0.permute(0, 2, 1))
            f1 = self.gather_local_1(grouped)
        h = self.conv_fuse[2](conv1x1(self.conv_fuse[0], self.conv_fuse[1], torch.cat([self.pt_last(f1), f1], dim=1)))
        g = h.max(dim=2)[0] if h.is_cuda else F.adaptive_max_pool1d(h, 1).view(B, -1)
        g = self.dp1(F.leaky_relu(self.bn6(self.linear1(g)), negative_slope=0.2))
        g = self.dp2(F.leaky_relu(self.bn7(self.linear2(g)), negative_slope=0.2))
        return self.linear3(g)
