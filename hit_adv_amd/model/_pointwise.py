"""1x1 convolutions of the point-cloud victims as GEMMs.

MIOpen has no tuned solver for the shapes these networks feed to ``Conv1d(k=1)`` / ``Conv2d(1x1)`` (tens of thousands of
tiny "images": [B*npoint, C, nsample]); on gfx950 it falls back to ``naive_conv_*`` kernels -- 15 ms forward and 24 ms
backward PER CALL in PCT's Local_op (rocprofv3, profiles/), and ``MIOpenBatchNormFwdInferSpatialEst`` adds 1.5 ms per
BatchNorm.  A 1x1 convolution is a matrix product, so in eval mode on the GPU the layer is evaluated as one GEMM with
the BatchNorm folded into its weights; parameters, buffers and state_dict layout are untouched (the modules stay
``nn.Conv*`` / ``nn.BatchNorm*``), training mode and CPU tensors go through the modules themselves.
"""
import contextlib
import functools
import warnings
import weakref

import torch
import torch.nn.functional as F


# Per-layer constants of an attack (folded weights, their fp16 pieces), keyed WEAKLY by the conv module: an entry dies with
# its layer, so a new model that reuses a freed one's id() / storage addresses can never be served the old one's weights,
# and the cached GPU tensors are released with the model.  An entry also remembers WHICH BatchNorm it was folded with.
_FOLD_CACHE = weakref.WeakKeyDictionary()
_PIECE_CACHE = weakref.WeakKeyDictionary()
FP16_MAX = 65504.


def _versions(conv, bn):
    ts = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
    return tuple((t._version, t.data_ptr(), t._cdata) if t is not None else None for t in ts)


def invalidate_folded(model=None):
    """Forget the cached folded weights / fp16 pieces (of ``model``'s layers, or all).  Needed only after an edit the version
    counters cannot see -- writing through ``param.data`` --; ``load_state_dict``, optimiser steps and every in-place op on
    the parameter itself are noticed without it."""
    for cache in (_FOLD_CACHE, _PIECE_CACHE):
        if model is None:
            cache.clear()
        else:
            for m in model.modules():
                cache.pop(m, None)


def _folded(conv, bn):
    """(W [Cout,Cin], b [Cout] or None) of bn(conv(.)) in eval mode.  Unless ``WEIGHT_GRADS`` is set the pair is a constant of
    the attack: it is computed without autograd and CACHED per layer until one of its parameters / buffers changes (their
    version counters), so a forward pass does not spend half a dozen five-microsecond launches per layer on re-deriving it
    (PCT: ~100 of them per pass, 4 % of cfg5's kernel time in profiles/r03)."""
    if not WEIGHT_GRADS:
        ver = _versions(conv, bn)
        hit = _FOLD_CACHE.get(conv)
        if hit is not None and hit[0] == ver and (hit[1]() if hit[1] is not None else None) is bn:
            return hit[2], hit[3]
        with torch.no_grad():
            W, b = _fold_now(conv, bn)
            W = W.contiguous()
        if not (W.is_cuda and torch.cuda.is_current_stream_capturing()):  # a capture records, it does not compute: nothing to keep
            _FOLD_CACHE[conv] = (ver, weakref.ref(bn) if bn is not None else None, W, b)
        return W, b
    return _fold_now(conv, bn)


def _pieces(conv, W, kind, split):
    """The folded layer's fp16 pieces (forward operand, backward operand), split once per weight and kept with the layer.
    Returns None when a weight lies beyond fp16's range (or is not finite): the caller then takes its f32 path for this
    layer -- a property of the weights, decided here once, not a flag raised on every attack."""
    key = (kind, W.data_ptr(), W._version, W._cdata)
    hit = _PIECE_CACHE.get(conv)
    if hit is not None and hit[0] == key:
        return hit[1]
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing and not bool((W.abs().max() <= FP16_MAX).item()):  # NaN compares false: not representable either
        made = None
    else:
        made = (split(W), split(W.t().contiguous()))
    if not capturing:
        _PIECE_CACHE[conv] = (key, made)
    return made


def _fold_now(conv, bn):
    W = conv.weight.reshape(conv.out_channels, -1)  # [Cout, Cin]
    b = conv.bias
    if bn is not None:
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        W = W * s[:, None]
        b = bn.bias - bn.running_mean * s if b is None else (b - bn.running_mean) * s + bn.bias
    return W, b


def _fast(conv, bn, x):
    return x.is_cuda and not conv.training and (bn is None or not bn.training)


def conv1x1(conv, bn, x):
    """bn(conv(x)) for channels-major x [B, Cin, ...]; ``bn`` may be None."""
    if not _fast(conv, bn, x):
        y = conv(x)
        return bn(y) if bn is not None else y
    W, b = _folded(conv, bn)
    shp = x.shape
    y = torch.matmul(W, x.reshape(shp[0], shp[1], -1))  # [B, Cout, L]
    if b is not None:
        y = y + b[None, :, None]
    return y.view(shp[0], -1, *shp[2:])


def linear_pm(conv, bn, x):
    """The same layer applied to points-major x [..., Cin] -> [..., Cout] (one GEMM, no permutes)."""
    W, b = _folded(conv, bn)
    return F.linear(x, W, b)


class _LinearReLU(torch.autograd.Function):
    """relu(x W^T + b) with the bias and the ReLU applied in the GEMM's epilogue (hipBLASLt through
    ``torch._addmm_activation``, which has no autograd formula of its own): the separate activation pass over the
    [points, Cout] tensor -- a read and a write as large as the GEMM's own output -- is gone."""

    @staticmethod
    def forward(ctx, x2, W, b):
        y = torch._addmm_activation(b, x2, W.t(), use_gelu=False)
        ctx.save_for_backward(x2, W, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x2, W, y = ctx.saved_tensors
        gm = torch.ops.aten.threshold_backward(g.contiguous(), y, 0)
        dx = gm @ W if ctx.needs_input_grad[0] else None
        dW = gm.t() @ x2 if ctx.needs_input_grad[1] else None
        db = gm.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dW, db


class _LinearReLUGatedLater(torch.autograd.Function):
    """``_LinearReLU`` whose consumer applies this layer's ReLU backward itself: the gradient arriving here is already gated by
    (y > 0) (``ops.group_linear_max(..., relu_input=True)`` does it on the way out of its backward kernel), so the backward
    pass is the GEMM alone -- the separate pass over the [rows, Cout] gradient and activation is gone."""

    @staticmethod
    def forward(ctx, x2, W, b):
        y = torch._addmm_activation(b, x2, W.t(), use_gelu=False)
        ctx.save_for_backward(W)
        return y

    @staticmethod
    def backward(ctx, g):
        W, = ctx.saved_tensors
        return g.contiguous() @ W, None, None


class _RowsLinearReLUGatedLater(torch.autograd.Function):
    """``_LinearReLUGatedLater`` on ``ops.rows_linear`` (csrc/rows_linear.hip): forward and input gradient of a narrow shared layer
    over 0.5-1 M grouped rows are bound by their reads and writes, which the library's f32 GEMM moves at a third of the HBM rate."""

    @staticmethod
    def forward(ctx, x2, pieces, b, flag):
        from .. import ops
        ctx.back = (pieces[1], flag)
        return ops.rows_linear(x2, pieces[0], b, True, flag)

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        Wt2, flag = ctx.back
        return ops.rows_linear(g.contiguous(), Wt2, None, False, flag), None, None, None


FUSED_ROWS_LINEAR = True  # the middle shared layer of a sample-and-group block as ops.rows_linear (fp16x2) where its widths allow
WEIGHT_GRADS = False  # True: the fused layers also return gradients for the (eval-mode) parameters


def linear_relu_pm(conv, bn, x):
    """relu(bn(conv(.))) applied to points-major x [..., Cin] -> [..., Cout].  A custom autograd node cannot tell
    whether a backward pass wants its weight gradient (``needs_input_grad`` only says the weights COULD receive one), so
    the folded weights enter as constants unless ``WEIGHT_GRADS`` is set: an attack differentiates with respect to the
    points, and the weight-gradient GEMM would cost as much as the input-gradient one."""
    W, b = _folded(conv, bn)
    if b is None or not x.is_cuda:
        return F.relu(F.linear(x, W, b))
    if not WEIGHT_GRADS:
        W, b = W.detach(), b.detach()
    y = _LinearReLU.apply(x.reshape(-1, x.shape[-1]), W, b)
    return y.view(*x.shape[:-1], W.shape[0])


_RANGE_FLAGS = {}


def range_flag(device):
    """One device-resident int32 per GPU that the fp16x2 kernels raise when an operand leaves fp16's range (65504)."""
    key = str(device)
    if key not in _RANGE_FLAGS:
        _RANGE_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _RANGE_FLAGS[key]


class Fp16RangeExceeded(RuntimeError):
    """An operand of an fp16x2 (two-piece fp16) layer lay beyond fp16's range (65504) or was not finite: what that pass
    computed is invalid.  The attacks catch it and run again in arithmetic with fp32's range (``degrade_on_fp16_range``)."""


def check_range(device):
    """Raise ``Fp16RangeExceeded`` if any fp16x2 layer of a victim has seen an operand beyond fp16's range since the flag
    was last cleared (one 4-byte read; the attacks call it where they read their results back anyway)."""
    flag = _RANGE_FLAGS.get(str(device))
    if flag is not None and int(flag.item()) != 0:
        flag.zero_()
        raise Fp16RangeExceeded("a fused linear + max / pooling layer (fp16x2 matrix form) met an activation beyond fp16's "
                                "range (65504) or a NaN: the results of this pass are invalid")


_FULL_RANGE_DEPTH = 0


@contextlib.contextmanager
def full_range_arithmetic():
    """Every fp16x2 form off for the duration: PointNet's shared layers as three bf16 pieces (fp32's range, fp32-accurate),
    the fused group / embedding layers of the other victims as their f32 GEMM compositions."""
    global FUSED_GROUP_MAX, FUSED_EMBEDDING_POOL, FUSED_ROWS_LINEAR, _FULL_RANGE_DEPTH
    from .dgcnn import FoldedDGCNN
    from .pointnet import FoldedPointNet
    saved = (FUSED_GROUP_MAX, FUSED_EMBEDDING_POOL, FoldedPointNet.matrix_mode, FoldedDGCNN.fused_embedding, FUSED_ROWS_LINEAR)
    FUSED_GROUP_MAX = FUSED_EMBEDDING_POOL = FoldedDGCNN.fused_embedding = FUSED_ROWS_LINEAR = False
    if FoldedPointNet.matrix_mode == 'fp16x2':
        FoldedPointNet.matrix_mode = 'bf16x3'
    _FULL_RANGE_DEPTH += 1
    try:
        yield
    finally:
        _FULL_RANGE_DEPTH -= 1
        FUSED_GROUP_MAX, FUSED_EMBEDDING_POOL, FoldedPointNet.matrix_mode, FoldedDGCNN.fused_embedding, FUSED_ROWS_LINEAR = saved


_DEGRADE_WARNED = False


def degrade_on_fp16_range(attack):
    """Decorator of an attack's entry point.  The reference never fails on range: when the fp16x2 layers of the victim report
    an operand beyond 65504 (``Fp16RangeExceeded``, raised where the attack reads its results back), the same call runs
    again under ``full_range_arithmetic`` -- the CPU generator rewound, so the second run takes the same draws, captures its
    own graphs and returns exactly what a process configured that way from the start returns.  One warning per process."""
    @functools.wraps(attack)
    def guarded(self, *args, **kwargs):
        global _DEGRADE_WARNED
        if _FULL_RANGE_DEPTH:  # already inside a degraded run (attack_many -> attack, subclass -> base)
            return attack(self, *args, **kwargs)
        rng = torch.get_rng_state()
        try:
            return attack(self, *args, **kwargs)
        except Fp16RangeExceeded as e:
            if not _DEGRADE_WARNED:
                _DEGRADE_WARNED = True
                warnings.warn("hit_adv_amd: %s -- running this attack again with the victim's fp16x2 layers in arithmetic of "
                              "fp32's range (PointNet: matrix_mode 'bf16x3'; the others: f32 GEMMs); further occurrences are "
                              "handled the same way without a warning" % (e,), RuntimeWarning, stacklevel=2)
            torch.set_rng_state(rng)
            with full_range_arithmetic():
                return attack(self, *args, **kwargs)
    return guarded


FUSED_GROUP_MAX = True  # last shared layer of a sample-and-group block + max over the neighbours as one fp16x2 MFMA kernel


def _fused_group_max(conv, W, b, x):
    """Which fused form of "last shared layer + max over the neighbours" applies to this layer and input, with its fp16
    pieces: ('reg' | 'g16', pieces, flag), or None (unsupported widths, CPU, weight gradients wanted, or weights beyond
    fp16's range: the f32 GEMM + max then)."""
    from .. import ops
    if not (FUSED_GROUP_MAX and x.is_cuda and b is not None and not WEIGHT_GRADS):
        return None
    flag = range_flag(x.device)
    if ops.group_linear_max_supported(W.shape[1], W.shape[0], x.shape[-2]):
        kind, pieces = 'reg', _pieces(conv, W, 'reg', lambda M: ops.split_weights_f16x2(M, range_flag=flag))
    elif ops.group_linear_max_g16_supported(W.shape[1], W.shape[0], x.shape[-2]):
        # widths the register-resident kernels do not cover (PCT's second Local_op, 256 -> 256): the tiled GEMM core
        kind, pieces = 'g16', _pieces(conv, W, 'g16', lambda M: ops.split_rows_f16x2(M, flag))
    else:
        return None
    return None if pieces is None else (kind, pieces, flag)


def linear_relu_then_max_pm(conv_mid, bn_mid, conv, bn, x):
    """Two shared layers and the max over the neighbours: relu(bn_mid(conv_mid(x))) -> ``linear_relu_max_pm``.  Where the fused
    last layer applies, the middle layer's ReLU backward rides on that kernel's output (no separate pass over the
    [.., ns, C] gradient); otherwise the plain composition."""
    Wm, bm = _folded(conv_mid, bn_mid)
    W, b = _folded(conv, bn)
    if bm is not None and _fused_group_max(conv, W, b, x) is not None:  # (only x's device and ns matter there)
        from .. import ops
        x2 = x.reshape(-1, x.shape[-1])
        pieces = None
        if FUSED_ROWS_LINEAR and ops.rows_linear_supported(Wm.shape[1], Wm.shape[0]):
            flag = range_flag(x.device)
            pieces = _pieces(conv_mid, Wm, 'reg', lambda M: ops.split_weights_f16x2(M, range_flag=flag))
        if pieces is not None:
            y = _RowsLinearReLUGatedLater.apply(x2.contiguous(), pieces, bm.detach(), flag)
        else:
            y = _LinearReLUGatedLater.apply(x2, Wm.detach(), bm.detach())
        return linear_relu_max_pm(conv, bn, y.view(*x.shape[:-1], Wm.shape[0]), relu_input=True)
    return linear_relu_max_pm(conv, bn, linear_relu_pm(conv_mid, bn_mid, x))


def grouped_first_two_then_max_pm(U, V, idx, conv_mid, bn_mid, conv, bn):
    """A sample-and-group block behind its first layer's two per-point products: relu(U[idx] + V) -> middle shared layer -> last
    shared layer + max over the neighbours.  Where the widths allow, the gather / add / ReLU runs inside the middle layer's kernel
    (``ops.GroupAddReLULinear``); otherwise ``ops.group_add_relu`` and ``linear_relu_then_max_pm``."""
    from .. import ops
    Wm, bm = _folded(conv_mid, bn_mid)
    W, b = _folded(conv, bn)
    B, S, ns = idx.shape
    if (FUSED_ROWS_LINEAR and U.is_cuda and bm is not None and not WEIGHT_GRADS
            and ops.group_add_relu_linear_supported(Wm.shape[1], Wm.shape[0], S, ns)):
        probe = torch.empty(0, ns, Wm.shape[0], device=U.device)  # (only the device and ns matter to the last layer's choice)
        if _fused_group_max(conv, W, b, probe) is not None:
            flag = range_flag(U.device)
            pieces = _pieces(conv_mid, Wm, 'reg', lambda M: ops.split_weights_f16x2(M, range_flag=flag))
            if pieces is not None:
                y = ops.GroupAddReLULinear.apply(U, V, idx, pieces[0], pieces[1], bm.detach(), flag)
                return linear_relu_max_pm(conv, bn, y, relu_input=True)
    return linear_relu_then_max_pm(conv_mid, bn_mid, conv, bn, ops.group_add_relu(U, V, idx))


def linear_relu_max_pm(conv, bn, x, relu_input=False):
    """relu(bn(conv(.))) applied to points-major x [..., ns, Cin] followed by the max over the ns neighbours -> [..., Cout]:
    ``hitadv_group_linear_max`` where the shape is supported (no [.., ns, Cout] activation, no ReLU / max passes, a sparse
    backward), the GEMM + max otherwise.  ``relu_input``: see ``ops.GroupLinearMax`` (fused path only)."""
    from .. import ops
    W, b = _folded(conv, bn)
    fused = _fused_group_max(conv, W, b, x)
    if fused is not None and fused[0] == 'reg':
        return ops.group_linear_max(x.contiguous(), W, b, fused[2], pieces=fused[1], relu_input=relu_input)
    if fused is not None:
        return ops.group_linear_max_g16(x.contiguous(), fused[1][0], fused[1][1], b, fused[2], relu_input=relu_input)
    assert not relu_input, "relu_input is a property of the fused paths"
    return linear_relu_pm(conv, bn, x).max(dim=-2)[0]


FUSED_EMBEDDING_POOL = True  # a wide layer + LeakyReLU + pooling over the points as one fp16x2 GEMM kernel (csrc/gemm16.hip)


def linear_lrelu_maxpool_pm(conv, bn, x, slope=0.2):
    """max over the points of lrelu(bn(conv(x))) for points-major x [B,n,Cin] -> [B,Cout] (PCT's ``conv_fuse`` + LeakyReLU +
    ``adaptive_max_pool1d``, model/pct_cls.py:65-68): ``hitadv_linear_lrelu_pool`` where the widths allow -- the [B,n,Cout]
    activation never exists, forward or backward --, else the GEMM + ``ops.lrelu_pool``."""
    from .. import ops
    W, b = _folded(conv, bn)
    B, n, Cin = x.shape
    C = W.shape[0]
    if (FUSED_EMBEDDING_POOL and x.is_cuda and b is not None and not WEIGHT_GRADS and ops.gemm_f16x2_supported(C, Cin)
            and ops.gemm_f16x2_supported(Cin, C)):
        flag = range_flag(x.device)
        pieces = _pieces(conv, W, 'pool', lambda M: ops.split_rows_f16x2(M, flag))
        if pieces is not None:
            return ops.linear_lrelu_pool(x.reshape(B * n, Cin), pieces[0], pieces[1], b, B, n, slope, flag)[:, :C]
    z = F.linear(x, W, b)
    if x.is_cuda and ops.lrelu_pool_supported(C):
        return ops.lrelu_pool(z.contiguous(), slope)[:, :C]
    return F.leaky_relu(z, negative_slope=slope).max(dim=1)[0]


def split_first_layer(conv, bn, n_rel):
    """The first shared layer of a sample-and-group block, W [rel ; rest] + t with rel = (neighbour - centre)[:n_rel],
    as the pair (W, t) of the folded layer: the caller forms U = [x_j ; rest_j] W^T per POINT and V = -c_i W[:, :n_rel]^T
    + t (PointNet++) or V = c_i (W_centre - W_rel)^T + t (PCT) per CENTRE, and ``ops.group_add_relu`` does the rest.
    Constants unless ``WEIGHT_GRADS`` (see linear_relu_pm)."""
    W, b = _folded(conv, bn)
    if b is None:
        b = torch.zeros(W.shape[0], device=W.device, dtype=W.dtype)
    if not WEIGHT_GRADS:
        W, b = W.detach(), b.detach()
    return W, b


def fast_pm(conv, bn, x):
    """Whether the points-major fast path applies (eval mode, CUDA)."""
    return _fast(conv, bn, x)
