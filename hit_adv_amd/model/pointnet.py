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


# --------------------------------------------------------------------------------------------
# Attack-time view: the same function, laid out for the GPU.
# --------------------------------------------------------------------------------------------
def _fold(lin_w, lin_b, bn):
    """Fold an eval-mode BatchNorm1d into the preceding 1x1 conv / linear layer:
    bn(Wx + b) = (s*W) x + (s*(b - mean) + beta),  s = gamma / sqrt(var + eps).  Returns (W'^T, b')."""
    w = lin_w.reshape(lin_w.shape[0], -1)
    if bn is None:
        return w.t().contiguous(), lin_b.clone()
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return (w * s[:, None]).t().contiguous(), (lin_b - bn.running_mean) * s + bn.bias


class _LinearMaxOverPoints(torch.autograd.Function):
    """g[b,c] = max_n act(x[b,n,:] @ Wt[:,c] + bias[c])  with act = identity or ReLU, for x [B*N,Cin].

    Same value and same gradient as ``addmm -> (relu) -> view(B,N,C).max(1)`` under autograd, but
      * ReLU is applied to the [B,C] maxima (max and ReLU commute), not to the [B*N,C] activations;
      * the 134 MB activation is not kept for the backward: only the arg-max table [B,C] is;
      * the backward uses the fact that d/dy is non-zero at ONE point per (cloud, channel): instead of
        zero-filling and scattering a [B*N,C] gradient and running a dense [B*N,C]x[C,Cin] GEMM, it
        sums the B*C rows  dg[b,c] * W[c,:]  into their points with ``hitadv_linear_max_bwd`` (one wave
        per destination point, ascending c, no atomics -> bitwise reproducible).
    CUDA only; ``FoldedPointNet`` uses the plain formulation for CPU tensors."""

    @staticmethod
    def forward(ctx, x, Wt, W, bias, B, N, relu):
        from .. import ops
        if ops.linear_max_fwd_supported(*Wt.shape):
            # GEMM on the f32 matrix cores with the max/arg-max in its epilogue: the 134 MB activation never exists
            g, idx = ops.linear_max_fwd(x, Wt, B, N, bias=bias, relu=relu)
        else:
            y = torch.mm(x, Wt)
            # one read at HBM rate (torch's dim-1 max is ~2.5x slower); bias add and ReLU ride in the merge pass
            g, idx = ops.max_over_points(y, B, N, bias=bias, relu=relu)
            del y
        ctx.save_for_backward(W, idx, g if relu else None)
        ctx.dims = (B, N)
        return g

    @staticmethod
    def backward(ctx, dg):
        from .. import ops
        W, idx, act_out = ctx.saved_tensors
        B, N = ctx.dims
        return ops.linear_max_bwd(dg, W, idx, N, act_out), None, None, None, None, None, None


class _LinearReLU(torch.autograd.Function):
    """relu(x @ Wt + bias) with bias and ReLU in the GEMM epilogue (hipBLASLt, ``torch._addmm_activation``);
    backward = threshold on the saved output + one GEMM for dX (weights are constants of the attack)."""

    @staticmethod
    def forward(ctx, x, Wt, W, bias):
        y = torch._addmm_activation(bias, x, Wt, use_gelu=False)
        ctx.save_for_backward(y, W)
        return y

    @staticmethod
    def backward(ctx, g):
        y, W = ctx.saved_tensors
        return torch.mm(torch.ops.aten.threshold_backward(g, y, 0), W), None, None, None


class _PointNetHip(torch.autograd.Function):
    """logits, trans_feat = f(x) and d/dx of it, entirely on libhitadv_hip (csrc/pointnet.hip + the fused 128->1024
    layers of csrc/victim_bf3.hip / victim.hip, in the view's ``matrix_mode``): 15 launches forward, 14 backward, no
    rocBLAS/MIOpen.
    Weights are constants (no weight gradients: the attack never uses them)."""

    @staticmethod
    def forward(ctx, x, v):
        from .. import ops
        B, _, N = x.shape
        R = B * N
        E = lambda *s: torch.empty(*s, device=x.device)  # noqa: E731
        # the shared-layer chains: on the fp16 matrix cores with the 128-wide activations handed on as packed pieces, or f32 MFMA
        cm, rf = (2, v.range_flag) if v.matrix_mode == 'fp16x2' else (0, None)
        # STN3d
        a1s, a2s = E(R, 64), E(R, 128)
        def lin_max(a, name, relu):
            """128 -> 1024 shared layer + max over the points in the view's matrix mode."""
            if v.matrix_mode == 'bf16x3':
                return ops.linear_max_fwd_bf16x3(a, v.pieces(name), B, N, bias=getattr(v, name + '_b'), relu=relu,
                                                 blocks=v.linear_max_blocks)
            if v.matrix_mode == 'fp16x2':
                return ops.linear_max_fwd_f16x2(a, v.pieces(name, 2), B, N, bias=getattr(v, name + '_b'), relu=relu,
                                                blocks=v.linear_max_blocks, range_flag=v.range_flag, packed=True)
            return ops.linear_max_fwd(a, getattr(v, name + '_w'), B, N, bias=getattr(v, name + '_b'), relu=relu)

        if v.deform_inputs is not None:  # x is an OUTPUT of the first kernel: the caller's deformation, evaluated inside
            ori, central, P, sigma, inv_den = v.deform_inputs
            v.deform_inputs = None  # consumed: the caller checks this
            ops.pointnet_rowmlp_fwd_deform(B, N, ori, central, P, sigma, x, inv_den, v.s1_w, v.s1_b, v.s2_w, v.s2_b, a1s, a2s,
                                           mode=cm, range_flag=rf)
        else:
            ops.pointnet_rowmlp_fwd(0, B, N, v.s2_w, v.s2_b, a2s, x=x, W0=v.s1_w, b0=v.s1_b, o0=a1s, mode=cm, range_flag=rf)
        gs, js = lin_max(a2s, 's3', True)
        f4s = ops.fc_layer(gs, v.s4_w, v.s4_b, relu=True)
        f5s = ops.fc_layer(f4s, v.s5_w, v.s5_b, relu=True)
        # input transform (STN3d's last layer is evaluated inside the stage-1 kernel), first encoder layer, STNkd
        h1, a1t, a2t = E(R, 64), E(R, 64), E(R, 128)
        if v.fold_small_layers:
            T3 = E(B, 9)
            ops.pointnet_rowmlp_fwd_stn(B, N, x, f5s, v.s6_w, v.s6_b, T3, v.e1_w, v.e1_b, v.t1_w, v.t1_b, v.t2_w, v.t2_b,
                                        h1, a1t, a2t, mode=cm, range_flag=rf)
        else:
            T3 = ops.fc_layer(f5s, v.s6_w, v.s6_b)
            ops.pointnet_rowmlp_fwd(1, B, N, v.t2_w, v.t2_b, a2t, x=x, T=T3, W0=v.e1_w, b0=v.e1_b, W1=v.t1_w, b1=v.t1_b,
                                    o0=h1, o1=a1t, mode=cm, range_flag=rf)
        gt, jt = lin_max(a2t, 't3', True)
        f4t = ops.fc_layer(gt, v.t4_w, v.t4_b, relu=True)
        f5t = ops.fc_layer(f4t, v.t5_w, v.t5_b, relu=True)
        T64 = ops.fc_layer(f5t, v.t6_w, v.t6_b)
        # feature transform, encoder tail, classifier head
        a2e = E(R, 128)
        ops.pointnet_rowmlp_fwd(2, B, N, v.e2_w, v.e2_b, a2e, T=T64, hin=h1, mode=cm, range_flag=rf)
        g, je = lin_max(a2e, 'e3', False)
        f1 = ops.fc_layer(g, v.h1_w, v.h1_b, relu=True)
        f2 = ops.fc_layer(f1, v.h2_w, v.h2_b, relu=True)
        if v.defer_logits:  # the caller's loss kernel evaluates the last layer itself (hitadv_iteration_head_reg) and
            logits = E(B, v.h3_w.shape[1])  # fills this buffer; nothing else may read it before that
            v.pending_head = (f2, v.h3_w, v.h3_b)
        else:
            logits = ops.fc_layer(f2, v.h3_w, v.h3_b)
        ctx.save_for_backward(x, a1s, a2s, gs, js, f4s, f5s, T3, h1, a1t, a2t, gt, jt, f4t, f5t, T64, a2e, je, f1, f2)
        ctx.view = v
        ctx.set_materialize_grads(False)
        return logits, T64.view(B, 64, 64)

    @staticmethod
    def backward(ctx, dlogits, dT64_ext):
        from .. import ops
        x, a1s, a2s, gs, js, f4s, f5s, T3, h1, a1t, a2t, gt, jt, f4t, f5t, T64, a2e, je, f1, f2 = ctx.saved_tensors
        v = ctx.view
        B, _, N = x.shape
        R = B * N
        E = lambda *s: torch.empty(*s, device=x.device)  # noqa: E731
        cm = 2 if v.matrix_mode == 'fp16x2' else 0  # the mode the forward pass ran in (2: a2* hold packed pieces)
        tiles, words = ops.pointnet_rowmlp_bwd_tiles(B, N, cm)
        if dlogits is None:
            dlogits = torch.zeros(B, v.h3_w.shape[1], device=x.device)
        # head and encoder tail
        if v.fold_small_layers:  # fc3 and fc2 backwards: one launch
            d = ops.fc_layer_pre(dlogits.contiguous().unsqueeze(1), v.h3_wr, v.h2_wr, mask=f2)
        else:
            d = ops.fc_layer(ops.fc_layer(dlogits.contiguous(), v.h3_wr), v.h2_wr, mask=f2)
        dg = ops.fc_layer(d, v.h1_wr, mask=f1)
        dTp, dH1 = E(B, tiles, 4096), E(R, 64)
        # which points of a tile receive any gradient: handed from stage to stage, each stage works on those rows only
        pres2 = torch.empty(B, tiles, words, device=x.device, dtype=torch.int64)
        pres1 = torch.empty(B, tiles, words, device=x.device, dtype=torch.int64)
        over = torch.empty(B, tiles, device=x.device, dtype=torch.int32) if words > 1 else None  # tiles left to the second launch
        ops.pointnet_rowmlp_bwd(2, B, N, dg, je, v.e3_wr, a2e, v.e2_wr, dH1, H1=h1, T=T64, dTpart=dTp, pres_out=pres2, mode=cm, overflow=over, words=words)
        dT64 = ops.sum_partials(dTp, None if dT64_ext is None else dT64_ext.reshape(B, 4096).contiguous())
        # STNkd, first encoder layer, input transform
        d = ops.fc_layer(dT64, v.t6_wr)
        d = ops.fc_layer(d, v.t5_wr, mask=f5t)
        dgt = ops.fc_layer(d, v.t4_wr, mask=f4t)
        dTp, dPts = E(B, tiles, 9), E(B, 3, N)
        ops.pointnet_rowmlp_bwd(1, B, N, dgt, jt, v.t3_wr, a2t, v.t2_wr, dPts, gmask=gt, A1=a1t, W1r=v.t1_wr, H1=h1,
                                dH1in=dH1, W0r=v.e1_wr, T=T3, x=x, dTpart=dTp, pres_in=pres2, pres_out=pres1, mode=cm, overflow=over, words=words)
        # STN3d: the sum of the tiles' dT3 partials, fc3 and fc2 backwards in one launch
        if v.fold_small_layers:
            d = ops.fc_layer_pre(dTp, v.s6_wr, v.s5_wr, mask=f5s)
        else:
            d = ops.fc_layer(ops.fc_layer(ops.sum_partials(dTp), v.s6_wr), v.s5_wr, mask=f5s)
        dgs = ops.fc_layer(d, v.s4_wr, mask=f4s)
        dX = E(B, 3, N)
        ops.pointnet_rowmlp_bwd(0, B, N, dgs, js, v.s3_wr, a2s, v.s2_wr, dX, gmask=gs, A1=a1s, W0r=v.s1_wr, dPin=dPts,
                                pres_in=pres1, mode=cm, overflow=over, words=words)
        return dX, None


class FoldedPointNet(nn.Module):
    """Inference-mode restatement of ``PointNetFeatureModel`` for the attack loop (still plain
    PyTorch-ROCm ops: rocBLAS/hipBLASLt GEMMs + elementwise).  Algebraically identical to the module
    in eval mode; what changes is the execution plan:

    * every BatchNorm is folded into the layer in front of it (weights are frozen during an attack);
    * activations are kept points-major ``[B*N, C]`` so each shared layer is ONE ``addmm`` with the
      bias in the GEMM epilogue, the two learned transforms are natural ``bmm``s and no transposes
      or ``contiguous()`` copies of the 134 MB activations are needed;
    * ReLU is applied in place;
    * the 128->1024 layer + max over points of each stack is one autograd node whose backward exploits
      that the max routes gradient to one point per (cloud, channel) (``_LinearMaxOverPoints``).

    This cuts the victim's forward+backward from ~250 to ~90 kernels per attack iteration.  Build it
    with ``PointNetFeatureModel.attack_view()``; it snapshots the weights at that moment.
    """

    PIECED = ('s3', 't3', 'e3')  # the 128 -> 1024 layers whose weights are also kept as three bf16 pieces

    def __init__(self, m):
        super().__init__()
        for k, (w, b) in self._folded(m).items():
            self.register_buffer(k + '_w', w.detach().clone())
            self.register_buffer(k + '_b', b.detach().clone())
            self.register_buffer(k + '_wr', w.detach().t().contiguous())  # row-major [Cout,Cin] for dX
        for name in self.PIECED:  # registered (non-persistent) buffers: .to() / .cuda() move them with the weights
            wr = getattr(self, name + '_wr')
            self.register_buffer(name + '_w3', torch.zeros(3, *wr.shape, dtype=torch.int16, device=wr.device), persistent=False)
            self.register_buffer(name + '_w2', torch.zeros(2, *wr.shape, dtype=torch.int16, device=wr.device), persistent=False)
        # raised by the fp16x2 kernels when a weight or an activation lies beyond fp16's range (65504)
        self.register_buffer('range_flag', torch.zeros(1, dtype=torch.int32, device=self.s3_wr.device), persistent=False)
        self._split_on = None  # the device the pieces were last split on (the split itself is a HIP kernel)
        if self.s3_wr.is_cuda:
            self._resplit()

    @staticmethod
    def _folded(m):
        assert not m.training, "attack_view() folds running statistics: call model.eval() first"
        f = m.feat
        if f.conv1.in_channels != 3 or not f.feature_transform:
            raise NotImplementedError("attack_view: xyz-only input with feature transform (the eval.py victim)")
        with torch.no_grad():
            layers = {
                's1': _fold(f.stn.conv1.weight, f.stn.conv1.bias, f.stn.bn1),
                's2': _fold(f.stn.conv2.weight, f.stn.conv2.bias, f.stn.bn2),
                's3': _fold(f.stn.conv3.weight, f.stn.conv3.bias, f.stn.bn3),
                's4': _fold(f.stn.fc1.weight, f.stn.fc1.bias, f.stn.bn4),
                's5': _fold(f.stn.fc2.weight, f.stn.fc2.bias, f.stn.bn5),
                's6': _fold(f.stn.fc3.weight, f.stn.fc3.bias + torch.eye(3, device=f.stn.fc3.bias.device).reshape(-1), None),
                'e1': _fold(f.conv1.weight, f.conv1.bias, f.bn1),
                't1': _fold(f.fstn.conv1.weight, f.fstn.conv1.bias, f.fstn.bn1),
                't2': _fold(f.fstn.conv2.weight, f.fstn.conv2.bias, f.fstn.bn2),
                't3': _fold(f.fstn.conv3.weight, f.fstn.conv3.bias, f.fstn.bn3),
                't4': _fold(f.fstn.fc1.weight, f.fstn.fc1.bias, f.fstn.bn4),
                't5': _fold(f.fstn.fc2.weight, f.fstn.fc2.bias, f.fstn.bn5),
                't6': _fold(f.fstn.fc3.weight, f.fstn.fc3.bias + torch.eye(64, device=f.fstn.fc3.bias.device).reshape(-1), None),
                'e2': _fold(f.conv2.weight, f.conv2.bias, f.bn2),
                'e3': _fold(f.conv3.weight, f.conv3.bias, f.bn3),
                'h1': _fold(m.fc1.weight, m.fc1.bias, m.bn1),
                'h2': _fold(m.fc2.weight, m.fc2.bias, m.bn2),
                'h3': _fold(m.fc3.weight, m.fc3.bias, None),
            }
        return layers

    def refresh(self, m):
        """Re-fold the module's current weights into the existing buffers (addresses stay fixed, so a
        captured hipGraph keeps working after the victim's weights change)."""
        for k, (w, b) in self._folded(m).items():
            getattr(self, k + '_w').copy_(w)
            getattr(self, k + '_b').copy_(b)
            getattr(self, k + '_wr').copy_(w.t())
        if self.s3_wr.is_cuda:
            self._resplit()  # in place: the addresses a captured graph holds stay valid
        return self

    def _resplit(self):
        from .. import ops
        self.range_flag.zero_()
        for name in self.PIECED:
            ops.split_weights_bf16x3(getattr(self, name + '_wr'), out=getattr(self, name + '_w3'))
            ops.split_weights_f16x2(getattr(self, name + '_wr'), out=getattr(self, name + '_w2'), range_flag=self.range_flag)
        self._split_on = self.s3_wr.device

    def check_range(self):
        """fp16x2 mode: raise ``Fp16RangeExceeded`` if any operand of the shared layers left fp16's range since the last
        refresh (one small device-to-host read; the attacks call it where they read their results back anyway, and run
        again in 'bf16x3' when it fires: ``_pointwise.degrade_on_fp16_range``)."""
        if self.matrix_mode == 'fp16x2' and self.range_flag.is_cuda and int(self.range_flag.item()) != 0:
            from ._pointwise import Fp16RangeExceeded
            raise Fp16RangeExceeded("FoldedPointNet(matrix_mode='fp16x2'): an activation or weight of a shared layer exceeds "
                                    "fp16's range (65504) or is not finite; the results of this pass are invalid "
                                    "(matrix_mode='bf16x3' and 'f32' have fp32's range)")

    linear_max_blocks = 0     # workgroups of the bf16x3 128 -> 1024 kernel (0 = one per CU); HiT_ADV.attack_many sets 128 on
    #                           ITS view while three or more attacks are in flight.  Per view, not per process.

    matrix_mode = 'fp16x2'  # the three 128 -> 1024 layers: 'bf16x3' (three bf16 pieces, six products: fp32-accurate), 'fp16x2' (two
    #                         fp16 pieces, three products: errors at fp32's unit roundoff, half the matrix time) or 'f32'
    deform_inputs = None      # (ori, central, perturb, sigma, inv_den) set by HiT-ADV's loop for ONE forward call: the input
    #                           tensor is then produced by the engine's first kernel (hitadv_pointnet_rowmlp_fwd_deform)
    defer_logits = False      # set by a caller whose loss kernel takes (features, last layer) instead of logits: see forward
    pending_head = None
    fold_small_layers = True  # the 256 -> 9 layer inside the stage-1 kernel, the 9 / 40 -> 256 backward layers inside the
    #                           next layer's launch (False: one launch per layer, kept for A/B timing and as a cross-check)

    def pieces(self, name, n=3):
        """bf16 pieces [3,Cout,Cin] (n = 3) or fp16 pieces [2,Cout,Cin] (n = 2) of a 128 -> 1024 layer's folded weight (weights are constants of an attack; ``refresh``
        re-splits them in place).  A view that was built on the CPU, or moved to another device since, is split here --
        eagerly at the first forward pass on the new device, never lazily inside someone's graph capture of a later one."""
        if self._split_on != self.s3_wr.device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FoldedPointNet: the bf16 weight pieces are stale (the view moved to %s); run one forward "
                                   "pass or call refresh() before capturing" % self.s3_wr.device)
            self._resplit()
        return getattr(self, name + ('_w3' if n == 3 else '_w2'))

    def _lin(self, x, name, relu=True):
        w, b = getattr(self, name + '_w'), getattr(self, name + '_b')
        if relu and x.is_cuda and x.shape[1] % 4 == 0:
            return _LinearReLU.apply(x, w, getattr(self, name + '_wr'), b)
        y = torch.addmm(b, x, w)
        return y.relu_() if relu else y

    def _lin_max(self, x, name, B, N, relu):
        """Last shared layer of a stack fused with the max over points (see _LinearMaxOverPoints)."""
        w, b = getattr(self, name + '_w'), getattr(self, name + '_b')
        if x.is_cuda:
            return _LinearMaxOverPoints.apply(x, w, getattr(self, name + '_wr'), b, B, N, relu)
        return self._lin(x, name, relu).view(B, N, -1).max(dim=1)[0]

    def _tnet(self, x, B, N, p):
        g = self._lin_max(self._lin(self._lin(x, p + '1'), p + '2'), p + '3', B, N, True)
        return self._lin(self._lin(self._lin(g, p + '4'), p + '5'), p + '6', relu=False)

    hip_engine = True  # CUDA tensors: run on libhitadv_hip's own kernels (_PointNetHip); False = PyTorch-ROCm ops
    iterations_per_graph = 10  # attack iterations an attack records into one hipGraph (~40 kernels each: launch-cost bound)

    def forward(self, x):
        """x [B,3,N] -> (logits [B,k], trans_feat [B,64,64])"""
        B, _, N = x.shape
        if x.is_cuda and self.hip_engine and x.dtype == torch.float32:
            return _PointNetHip.apply(x.contiguous(), self)
        pts = x.transpose(1, 2)  # [B,N,3] view
        trans = self._tnet(pts.reshape(B * N, 3), B, N, 's').view(B, 3, 3)
        h = self._lin(torch.bmm(pts, trans).reshape(B * N, 3), 'e1')  # [B*N,64]
        trans_feat = self._tnet(h, B, N, 't').view(B, 64, 64)
        h = torch.bmm(h.view(B, N, 64), trans_feat).reshape(B * N, 64)
        g = self._lin_max(self._lin(h, 'e2'), 'e3', B, N, False)
        return self._lin(self._lin(self._lin(g, 'h1'), 'h2'), 'h3', relu=False), trans_feat


def _attack_view(self):
    return FoldedPointNet(self)


PointNetFeatureModel.attack_view = _attack_view
