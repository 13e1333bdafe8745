"""Tensor-level front end of the HIP kernels (torch tensors in HBM -> C ABI calls).

torch is plumbing here: it owns device memory, the current HIP stream and the autograd
tape; every computation below happens in libhitadv_hip.so.  All functions require CUDA
(ROCm) tensors and raise otherwise -- there is no CPU path.
"""
import ctypes
import struct

import torch

from . import _lib

FORM_DIRECT, FORM_GRAM, FORM_GRAM_KNN, FORM_SQUARE_DISTANCE = 0, 1, 2, 3  # include/hitadv.h HITADV_FORM_*

_reference_arithmetic = False


class reference_arithmetic:
    """Switch (``reference_arithmetic.set(True)``) or scope (``with reference_arithmetic(True):``) in which the set
    distances and KNNDist evaluate squared distances in the reference's own Gram-form fp32 arithmetic
    (util/set_distance.py:15-32, util/dist_utils.py:148-150) instead of the direct form: their VALUES then equal the
    reference's bit for bit (minima, k-nearest distances; what remains is the order of the final mean), at the price of
    the Gram form's cancellation noise (~4 eps (|x|^2 + |y|^2) per entry).  Gradients flow through the same arg-minima /
    neighbour indices either way and are evaluated as 2 g (x - y)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _reference_arithmetic
        self.prev, _reference_arithmetic = _reference_arithmetic, self.on
        return self

    def __exit__(self, *exc):
        global _reference_arithmetic
        _reference_arithmetic = self.prev

    @staticmethod
    def set(on):
        global _reference_arithmetic
        _reference_arithmetic = bool(on)

    @staticmethod
    def get():
        return _reference_arithmetic


class victim_reference_arithmetic(reference_arithmetic):
    """The same kind of switch for the samplers INSIDE the victims (PointNet++'s ball query, PCT's farthest point
    sampling and kNN grouping): on (the default), their distances are evaluated in the reference's own fp32 arithmetic
    (Gram-form ``square_distance``, model/pointnet2_utils.py:19-41; ``get_dists``, util/other_utils.py:237-251), so the
    index tables -- which DEFINE the victim's function -- are the reference's bit for bit; off, in the direct form."""
    _on = True

    def __enter__(self):
        self.prev, victim_reference_arithmetic._on = victim_reference_arithmetic._on, self.on
        return self

    def __exit__(self, *exc):
        victim_reference_arithmetic._on = self.prev

    @staticmethod
    def set(on):
        victim_reference_arithmetic._on = bool(on)

    @staticmethod
    def get():
        return victim_reference_arithmetic._on


def _victim_form(explicit):
    return victim_reference_arithmetic._on if explicit is None else bool(explicit)


def _form(explicit, reference_form):
    if explicit is not None:
        return reference_form if explicit else FORM_DIRECT
    return reference_form if _reference_arithmetic else FORM_DIRECT


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _dev(t, name, dtype=torch.float32):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: hit_adv_amd has no CPU path" % name)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


_fc_scratch = {}


def _fc_scratch_for(x, n):
    """Zero-initialised ticket / split-K scratch, one per (device, stream): calls on one stream are ordered, calls on
    different streams (attack_many) must not share tickets.  Grown on demand, never shrunk.  Created on first use of a
    stream -- also inside a graph capture (the zero-fill then becomes a node of that graph, which is harmless: tickets
    are zero between launches anyway; the tensor keeps its block of the graph's pool alive)."""
    key = (x.device.index, torch.cuda.current_stream().cuda_stream)
    t = _fc_scratch.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 1 << 20), device=x.device)
        _fc_scratch[key] = t
    return t


# --------------------------------------------------------------------------- pairwise / set minima
def pairwise_sqdist(x, y, form=FORM_GRAM):
    """x[B,N,D], y[B,M,D] -> P[B,N,M] (no autograd; see NNMin for the differentiable reductions)."""
    x, y = _dev(x.detach(), "x"), _dev(y.detach(), "y")
    B, N, D = x.shape
    M = y.shape[1]
    P = torch.empty(B, N, M, device=x.device, dtype=torch.float32)
    _lib.call("hitadv_pairwise_sqdist", _p(x), _p(y), _p(P), B, N, M, D, form, _stream())
    return P


class NNMin(torch.autograd.Function):
    """(x[B,N,D], y[B,M,D]) -> (min_x[B,N], arg_x[B,N], min_y[B,M], arg_y[B,M])."""

    @staticmethod
    def forward(ctx, x, y, form=FORM_DIRECT):
        x, y = _dev(x, "x"), _dev(y, "y")
        B, N, D = x.shape
        M = y.shape[1]
        dev = x.device
        min_x = torch.empty(B, N, device=dev)
        min_y = torch.empty(B, M, device=dev)
        arg_x = torch.empty(B, N, device=dev, dtype=torch.int32)
        arg_y = torch.empty(B, M, device=dev, dtype=torch.int32)
        scratch = torch.empty(B, N, M, device=dev) if D != 3 else None
        if D != 3:
            form = FORM_DIRECT  # the generic-D path (quirk Q1's [B,3,N] call) has one form
        _lib.call("hitadv_nn_min", _p(x), _p(y), B, N, M, D, form, _p(min_x), _p(arg_x), _p(min_y), _p(arg_y),
                  _p(scratch), _stream())
        ctx.save_for_backward(x, y, arg_x, arg_y)
        ctx.mark_non_differentiable(arg_x, arg_y)
        return min_x, arg_x, min_y, arg_y

    @staticmethod
    def backward(ctx, g_min_x, _gax, g_min_y, _gay):
        x, y, arg_x, arg_y = ctx.saved_tensors
        B, N, D = x.shape
        M = y.shape[1]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None, None
        gmx = g_min_x.contiguous().float() if g_min_x is not None else None
        gmy = g_min_y.contiguous().float() if g_min_y is not None else None
        _lib.call("hitadv_nn_min_bwd", _p(x), _p(y), _p(arg_x), _p(arg_y), _p(gmx), _p(gmy), B, N, M, D,
                  _p(gx), _p(gy), _stream())
        return gx, gy, None


def nn_min(x, y, reference=None):
    """Fused nearest-neighbour minima in both directions.  ``reference`` = True / False selects the reference's Gram-form
    arithmetic or the direct form; None follows the ``reference_arithmetic`` switch."""
    return NNMin.apply(x, y, _form(reference, FORM_GRAM))


# --------------------------------------------------------------------------- kNN
class KnnPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, K, form=FORM_DIRECT):
        p1, p2 = _dev(p1, "p1"), _dev(p2, "p2")
        B, N, D = p1.shape
        M = p2.shape[1]
        if D != 3 or p2.shape[2] != 3:
            raise RuntimeError("knn_points: only 3-D points are supported by the HIP kernel")
        if not 1 <= K <= min(M, 64):
            raise RuntimeError("knn_points: need 1 <= K <= min(M, 64), got K=%d, M=%d" % (K, M))
        dists = torch.empty(B, N, K, device=p1.device)
        idx = torch.empty(B, N, K, device=p1.device, dtype=torch.int64)
        _lib.call("hitadv_knn_points", _p(p1), _p(p2), B, N, M, K, form, _p(dists), _p(idx), 1, _stream())
        ctx.save_for_backward(p1, p2, idx)
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    def backward(ctx, g_dists, _gi):
        p1, p2, idx = ctx.saved_tensors
        B, N, _ = p1.shape
        M = p2.shape[1]
        K = idx.shape[2]
        g1 = torch.empty_like(p1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(p2) if ctx.needs_input_grad[1] else None
        if g1 is None and g2 is None:
            return None, None, None, None
        g = g_dists.contiguous().float()
        _lib.call("hitadv_knn_points_bwd", _p(p1), _p(p2), _p(idx), 1, _p(g), B, N, M, K, _p(g1), _p(g2),
                  _stream())
        return g1, g2, None, None


def knn_dist_matrix_form(reference=None):
    """The distance form KNNDist asks of ``KnnPoints``: its own Gram matrix (util/dist_utils.py:148-150) or direct."""
    return _form(reference, FORM_GRAM_KNN)


# --------------------------------------------------------------------------- deformation
class Deform(torch.autograd.Function):
    """adv = ori + (sum_j k_j p_j) / (sum_j k_j),  k_j = exp(-|x - c_j| / (2 sigma_j^2)).

    Differentiable w.r.t. perturb[B,C,3] and sigma[B,C] only (ori / central are constants of the
    attack, ShapeAttack/HiT_ADV.py:57,63-93)."""

    @staticmethod
    def forward(ctx, ori, central, perturb, sigma):
        ori, central = _dev(ori, "ori"), _dev(central, "central")
        perturb, sigma = _dev(perturb, "perturb"), _dev(sigma, "sigma")
        B, _, N = ori.shape
        C = central.shape[2]
        adv = torch.empty_like(ori)
        inv_den = torch.empty(B, N, device=ori.device)
        _lib.call("hitadv_deform_fwd", _p(ori), _p(central), _p(perturb), _p(sigma), B, N, C, _p(adv),
                  _p(inv_den), _stream())
        ctx.save_for_backward(ori, central, perturb, sigma, adv, inv_den)
        return adv

    @staticmethod
    def backward(ctx, g_adv):
        ori, central, perturb, sigma, adv, inv_den = ctx.saved_tensors
        B, _, N = ori.shape
        C = central.shape[2]
        g_adv = g_adv.contiguous().float()
        n = _lib.load().hitadv_deform_bwd_scratch_floats(B, N, C)
        partials = torch.empty(n, device=ori.device)
        gp = torch.empty_like(perturb)
        gs = torch.empty_like(sigma)
        _lib.call("hitadv_deform_bwd", _p(ori), _p(central), _p(perturb), _p(sigma), _p(adv), _p(inv_den),
                  _p(g_adv), B, N, C, _p(partials), _p(gp), _p(gs), _stream())
        return None, None, gp, gs


def deform(ori, central, perturb, sigma):
    return Deform.apply(ori, central, perturb, sigma)


class Regulariser(torch.autograd.Function):
    """mean(scale_const) * (cd_w*Q1 + ker_w*(|P|+|1-sigma|)/C + hide_w*mean cos)  as ONE autograd node
    (three launches forward, one backward) instead of ~45 torch ops; see hitadv_regulariser_fwd."""

    @staticmethod
    def forward(ctx, perturb, sigma, adv, ori, hide_ref, scale_const, weights, sig_range, dist_out):
        perturb, sigma, adv = _dev(perturb, "perturb"), _dev(sigma, "sigma"), _dev(adv, "adv")
        B, _, N = adv.shape
        C = sigma.shape[1]
        scratch = torch.empty(_lib.load().hitadv_regulariser_scratch_floats(B), device=adv.device)
        scaled = torch.empty((), device=adv.device)
        cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
        lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
        _lib.call("hitadv_regulariser_fwd", _p(perturb), _p(sigma), _p(adv), _p(ori), _p(hide_ref), _p(scale_const),
                  B, N, C, cd, ker, hide, lo, hi, _p(scratch), _p(dist_out), _p(scaled), _stream())
        ctx.save_for_backward(perturb, sigma, adv, ori, hide_ref, scratch)
        ctx.cfg = (weights, sig_range)
        return scaled

    @staticmethod
    def backward(ctx, go):
        perturb, sigma, adv, ori, hide_ref, scratch = ctx.saved_tensors
        weights, sig_range = ctx.cfg
        B, _, N = adv.shape
        C = sigma.shape[1]
        gp, gs, ga = torch.empty_like(perturb), torch.empty_like(sigma), torch.empty_like(adv)
        cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
        lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
        go = go.contiguous().float()
        _lib.call("hitadv_regulariser_bwd", _p(perturb), _p(sigma), _p(adv), _p(ori), _p(hide_ref), _p(scratch),
                  _p(go), B, N, C, cd, ker, hide, lo, hi, _p(gp), _p(gs), _p(ga), _stream())
        return gp, gs, ga, None, None, None, None, None, None


def regulariser(perturb, sigma, adv, ori, hide_ref, scale_const, weights, sig_range, dist_out):
    return Regulariser.apply(perturb, sigma, adv, ori, hide_ref, scale_const, weights, sig_range, dist_out)


# --------------------------------------------------------------------------- the same pieces without autograd
# (explicit forward / backward calls on caller-owned buffers: what HiT_ADV._iteration_fused strings together)
def deform_fwd_into(ori, central, perturb, sigma, adv, inv_den):
    B, _, N = ori.shape
    _lib.call("hitadv_deform_fwd", _p(ori), _p(central), _p(perturb), _p(sigma), B, N, central.shape[2], _p(adv),
              _p(inv_den), _stream())


def deform_bwd_scratch(B, N, C):
    return int(_lib.load().hitadv_deform_bwd_scratch_floats(B, N, C))


def deform_bwd_into(ori, central, perturb, sigma, adv, inv_den, g_adv, partials, gp, gs):
    B, _, N = ori.shape
    _lib.call("hitadv_deform_bwd", _p(ori), _p(central), _p(perturb), _p(sigma), _p(adv), _p(inv_den), _p(g_adv), B, N,
              central.shape[2], _p(partials), _p(gp), _p(gs), _stream())


def regulariser_scratch(B):
    return int(_lib.load().hitadv_regulariser_scratch_floats(B))


def regulariser_fwd_into(perturb, sigma, adv, ori, hide_ref, scale_const, weights, sig_range, scratch, dist_out,
                         scaled_out):
    B, _, N = adv.shape
    cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
    lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
    _lib.call("hitadv_regulariser_fwd", _p(perturb), _p(sigma), _p(adv), _p(ori), _p(hide_ref), _p(scale_const), B, N,
              sigma.shape[1], cd, ker, hide, lo, hi, _p(scratch), _p(dist_out), _p(scaled_out), _stream())


def regulariser_bwd_add(perturb, sigma, adv, ori, hide_ref, scratch, add_adv, weights, sig_range, gp, gs, ga):
    """Gradients of the scaled regulariser (upstream 1) into gp / gs, and ga = its adv term + add_adv."""
    B, _, N = adv.shape
    cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
    lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
    _lib.call("hitadv_regulariser_bwd_add", _p(perturb), _p(sigma), _p(adv), _p(ori), _p(hide_ref), _p(scratch), None,
              _p(add_adv), B, N, sigma.shape[1], cd, ker, hide, lo, hi, _p(gp), _p(gs), _p(ga), _stream())


def adam_step_sum(perturb, sigma, gp, gp2, gs, gs2, m_p, v_p, m_s, v_s, step, lr_p, lr_s, clamp_p, clamp_s):
    """Adam on both groups with gradient g + g2 (g2 may be None) and the projection onto ``clamp_*`` afterwards;
    ``step`` (device int32[1]) already holds the 1-based step number."""
    _lib.call("hitadv_adam_step_sum", _p(perturb), _p(gp), _p(gp2), _p(m_p), _p(v_p), perturb.numel(),
              ctypes.c_float(lr_p), ctypes.c_float(clamp_p[0]), ctypes.c_float(clamp_p[1]), _p(sigma), _p(gs), _p(gs2),
              _p(m_s), _p(v_s), sigma.numel(), ctypes.c_float(lr_s), ctypes.c_float(clamp_s[0]),
              ctypes.c_float(clamp_s[1]), _p(step), _stream())


ADV_UNTARGETED, ADV_TARGETED, ADV_CROSS_ENTROPY = 0, 1, 2


def iteration_head_scratch(B, device):
    """Zeroed scratch of ``iteration_head`` (per-cloud loss terms + a ticket that every call leaves at zero)."""
    return torch.zeros(int(_lib.load().hitadv_iteration_head_scratch_floats(B)), device=device)


def iteration_head(logits, label, perturb, sigma, adv, state, counter, kind, kappa, loss_out, dlogits, scratch):
    """``best_update`` and ``adv_loss`` in one launch (same results): the best-so-far buffers of ``state`` are updated in
    place, ``loss_out`` (0-d) and ``dlogits`` [B,K] receive the adversarial loss and its gradient."""
    B, K = logits.shape
    _lib.call("hitadv_iteration_head", _p(logits), _p(label), _p(perturb), _p(sigma), _p(adv), B, K, adv.shape[2],
              sigma.shape[1], _p(state["bestdist"]), _p(state["bestscore"]), _p(state["o_bestdist"]),
              _p(state["o_bestscore"]), _p(state["o_bestattack"]), _p(state["pred"]), _p(state["dist_val"]), _p(counter),
              kind, ctypes.c_float(float(kappa)), _p(loss_out), _p(dlogits), _p(scratch), _stream())


def iteration_head_reg(logits, label, perturb, sigma, adv, state, counter, kind, kappa, loss_out, dlogits, scratch, ori,
                       hide_ref, scale_const, weights, sig_range, reg_scratch, dist_out, scaled_out, head=None, groups=1):
    """``iteration_head`` and ``regulariser_fwd_fused_into`` in one launch (same results, bit for bit).  ``head`` =
    (feat [B,F], Wt [F,K], bias [K]): ``logits`` is then an OUTPUT, the classifier's last layer is evaluated inside.
    ``groups`` = G > 1: every argument holds G stacked attacks (include/hitadv.h, ``*_stack``): per-cloud tensors [G*B, ...],
    per-group scalars [G], scratches [G, size]; each group gets the bits of a call of its own."""
    B, K = logits.shape[0] // groups, logits.shape[1]
    cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
    lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
    _lib.call(*(("hitadv_iteration_head_reg_stack", groups) if groups > 1 else ("hitadv_iteration_head_reg",)),
              _p(logits), _p(label), _p(perturb), _p(sigma), _p(adv), B, K, adv.shape[2],
              sigma.shape[1], _p(state["bestdist"]), _p(state["bestscore"]), _p(state["o_bestdist"]),
              _p(state["o_bestscore"]), _p(state["o_bestattack"]), _p(state["pred"]), _p(state["dist_val"]), _p(counter),
              kind, ctypes.c_float(float(kappa)), _p(loss_out), _p(dlogits), _p(scratch), _p(ori), _p(hide_ref),
              _p(scale_const), cd, ker, hide, lo, hi, _p(reg_scratch), _p(dist_out), _p(scaled_out),
              _p(head[0]) if head else None, _p(head[1]) if head else None, _p(head[2]) if head else None,
              int(head[0].shape[1]) if head else 0, _stream())


def regulariser_fwd_fused_into(perturb, sigma, adv, ori, hide_ref, scale_const, weights, sig_range, scratch, dist_out,
                               scaled_out):
    """``regulariser_fwd_into`` in one launch; ``scratch`` must have been ZEROED when it was allocated."""
    B, _, N = adv.shape
    cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
    lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
    _lib.call("hitadv_regulariser_fwd_fused", _p(perturb), _p(sigma), _p(adv), _p(ori), _p(hide_ref), _p(scale_const), B,
              N, sigma.shape[1], cd, ker, hide, lo, hi, _p(scratch), _p(dist_out), _p(scaled_out), _stream())


def deform_bwd_partials_into(ori, central, perturb, sigma, adv, inv_den, g_adv, partials):
    B, _, N = ori.shape
    _lib.call("hitadv_deform_bwd_partials", _p(ori), _p(central), _p(perturb), _p(sigma), _p(adv), _p(inv_den), _p(g_adv),
              B, N, central.shape[2], _p(partials), _stream())


def deform_bwd_partials_reg_into(ori, central, perturb, sigma, adv, inv_den, g_victim, reg_scratch, weights, partials, groups=1):
    """``deform_bwd_partials_into`` whose upstream gradient is ``g_victim`` plus the regularisers' term, evaluated inside
    from the forward pass's ``reg_scratch`` (what ``regulariser_bwd_add`` would have written to ``ga``).  ``groups``: see
    ``iteration_head_reg``."""
    B, N = ori.shape[0] // groups, ori.shape[2]
    _lib.call(*(("hitadv_deform_bwd_partials_reg_stack", groups) if groups > 1 else ("hitadv_deform_bwd_partials_reg",)),
              _p(ori), _p(central), _p(perturb), _p(sigma), _p(adv), _p(inv_den),
              _p(g_victim), _p(reg_scratch), ctypes.c_float(float(weights[0])), B, N, central.shape[2], _p(partials),
              _stream())


def deform_bwd_adam_reg(ori, central, perturb, sigma, adv, inv_den, g_victim, hide_ref, reg_scratch, weights, sig_range, m_p,
                        v_p, m_s, v_s, step, lr_p, lr_s, clamp_p, clamp_s, partials, tickets):
    """``deform_bwd_partials_reg_into`` and ``adam_step_partials_reg`` in one launch (same bits); ``tickets``: int32 [B],
    zeroed once."""
    B, _, N = ori.shape
    cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
    lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
    _lib.call("hitadv_deform_bwd_adam_reg", _p(ori), _p(central), _p(perturb), _p(sigma), _p(adv), _p(inv_den), _p(g_victim),
              _p(hide_ref), _p(reg_scratch), cd, ker, hide, lo, hi, _p(m_p), _p(v_p), _p(m_s), _p(v_s), B, N,
              central.shape[2], ctypes.c_float(lr_p), ctypes.c_float(clamp_p[0]), ctypes.c_float(clamp_p[1]),
              ctypes.c_float(lr_s), ctypes.c_float(clamp_s[0]), ctypes.c_float(clamp_s[1]), _p(step), _p(partials),
              _p(tickets), _stream())


def adam_step_partials_reg(perturb, sigma, partials, N, hide_ref, reg_scratch, weights, sig_range, m_p, v_p, m_s, v_s, step,
                           lr_p, lr_s, clamp_p, clamp_s, groups=1):
    """``adam_step_partials`` that evaluates the regularisers' gradients at (perturb, sigma) itself (``regulariser_bwd_add``'s
    ``gp`` / ``gs``) instead of reading them.  ``groups``: see ``iteration_head_reg``."""
    B, C = sigma.shape[0] // groups, sigma.shape[1]
    cd, ker, hide = (ctypes.c_float(float(w)) for w in weights)
    lo, hi = (ctypes.c_float(float(v)) for v in sig_range)
    _lib.call(*(("hitadv_adam_step_partials_reg_stack", groups) if groups > 1 else ("hitadv_adam_step_partials_reg",)),
              _p(perturb), _p(sigma), _p(partials),
              int(_lib.load().hitadv_deform_bwd_slabs(N)), _p(hide_ref), _p(reg_scratch), cd, ker, hide, lo, hi, _p(m_p),
              _p(v_p), _p(m_s), _p(v_s), B, C, ctypes.c_float(lr_p), ctypes.c_float(clamp_p[0]), ctypes.c_float(clamp_p[1]),
              ctypes.c_float(lr_s), ctypes.c_float(clamp_s[0]), ctypes.c_float(clamp_s[1]), _p(step), _stream())


def adam_step_partials(perturb, sigma, partials, N, gp2, gs2, m_p, v_p, m_s, v_s, step, lr_p, lr_s, clamp_p, clamp_s):
    """``adam_step_sum`` with the deformation's gradient still in its per-slab partials (summed inside, reduce order)."""
    B, C = sigma.shape
    _lib.call("hitadv_adam_step_partials", _p(perturb), _p(sigma), _p(partials), int(_lib.load().hitadv_deform_bwd_slabs(N)),
              _p(gp2), _p(gs2), _p(m_p), _p(v_p), _p(m_s), _p(v_s), B, C, ctypes.c_float(lr_p),
              ctypes.c_float(clamp_p[0]), ctypes.c_float(clamp_p[1]), ctypes.c_float(lr_s), ctypes.c_float(clamp_s[0]),
              ctypes.c_float(clamp_s[1]), _p(step), _stream())


def adv_loss(kind, logits, target, kappa=0., loss_out=None):
    """(loss, d loss / d logits) of the util/adv_utils.py losses in one launch; ``loss_out`` (0-d) is written if given."""
    logits = _dev(logits.detach(), "logits")
    target = _dev(target, "target", torch.int64)
    B, K = logits.shape
    loss = loss_out if loss_out is not None else torch.empty((), device=logits.device)
    d = torch.empty_like(logits)
    _lib.call("hitadv_adv_loss", kind, _p(logits), _p(target), B, K, ctypes.c_float(float(kappa)), _p(loss), _p(d),
              _stream())
    return loss, d


# --------------------------------------------------------------------------- attack state
def best_update(logits, label, perturb, sigma, adv, state, counter=None):
    """In-place update of the best-so-far buffers in ``state`` (see hitadv_best_update); ``counter`` (device
    int32[1], optional) is incremented once per call."""
    B, K = logits.shape
    N = adv.shape[2]
    C = sigma.shape[1]
    _lib.call("hitadv_best_update", _p(logits), _p(label), _p(perturb), _p(sigma), _p(adv), B, K, N, C,
              _p(state["bestdist"]), _p(state["bestscore"]), _p(state["o_bestdist"]), _p(state["o_bestscore"]),
              _p(state["o_bestattack"]), _p(state["pred"]), _p(state["dist_val"]),
              _p(counter if counter is not None else state.get("iter")), _stream())


def adam_step(perturb, sigma, g_perturb, g_sigma, m_p, v_p, m_s, v_s, step, lr_p, lr_s):
    """One torch.optim.Adam-equivalent step on both parameter groups; ``step`` is a device int32[1]."""
    g_perturb, g_sigma = g_perturb.contiguous(), g_sigma.contiguous()
    _lib.call("hitadv_adam_step", _p(perturb), _p(g_perturb), _p(m_p), _p(v_p), perturb.numel(),
              ctypes.c_float(lr_p), _p(sigma), _p(g_sigma), _p(m_s), _p(v_s), sigma.numel(),
              ctypes.c_float(lr_s), _p(step), _stream())


def adam_single(p, g, m, v, step, lr):
    """torch.optim.Adam's default update (betas 0.9/0.999, eps 1e-8, no weight decay) of ONE tensor in place; ``step`` is a
    device int32[1] that counts the updates."""
    g = g.contiguous()
    _lib.call("hitadv_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), ctypes.c_float(lr), None, None, None, None, 0,
              ctypes.c_float(0.), _p(step), _stream())


def assign(dst, src):
    """``dst.copy_(src)`` for the state an attack loop keeps at fixed addresses, as a KERNEL: a contiguous same-dtype
    ``copy_`` is hipMemcpyAsync, i.e. a memcpy node of the captured iteration, and those make graph launches on several
    streams serialise and hold the host (include/hitadv.h, hitadv_copy).  Anything else (dtype change, strides, broadcast,
    CPU) is an element-wise kernel already and goes through ``copy_``."""
    if (torch.is_tensor(src) and dst.is_cuda and src.is_cuda and dst.device == src.device and dst.dtype == src.dtype
            and dst.shape == src.shape and dst.is_contiguous() and src.is_contiguous()):
        _lib.call("hitadv_copy", _p(dst), _p(src), dst.numel() * dst.element_size(), _stream())
        torch.autograd.graph.increment_version(dst)  # an in-place write autograd did not see: saved-tensor checks stay honest
        return dst
    with torch.no_grad():
        return dst.copy_(src)


def copy_of(src):
    """``src.detach().clone()`` without the memcpy node (see ``assign``)."""
    src = src.detach()
    if src.is_cuda and src.is_contiguous():
        out = torch.empty_like(src)
        _lib.call("hitadv_copy", _p(out), _p(src), src.numel() * src.element_size(), _stream())
        return out
    return src.clone()


# --------------------------------------------------------------------------- FPS
def fps_from_start(xyz, npoint, start):
    """xyz[B,N,3], start[B] int64 -> idx[B,npoint] int64 (ShapeAttack/HiT_ADV.py:489-510 semantics)."""
    xyz = _dev(xyz.detach(), "xyz")
    start = _dev(start, "start", torch.int64)
    B, N, _ = xyz.shape
    idx = torch.empty(B, npoint, device=xyz.device, dtype=torch.int64)
    _lib.call("hitadv_fps_from_start", _p(xyz), _p(start), B, N, npoint, _p(idx), _stream())
    return idx


def fps_pct(xyz, npoint, start, reference=None):
    """PCT's sampler (util/other_utils.py:254-272) from given first indices: xyz[B,N,3], start[B] int64 -> idx[B,npoint]
    int64.  With ``reference`` on (None: the ``victim_reference_arithmetic`` switch) the running distances are the
    reference's sqrt(clamped Gram form) values bit for bit; off, squared direct-form distances (sqrt is monotone: the
    tables differ only where two candidates are within fp32 rounding of each other)."""
    if not _victim_form(reference):
        return fps_from_start(xyz, npoint, start)
    xyz = _dev(xyz.detach(), "xyz")
    start = _dev(start, "start", torch.int64)
    B, N, _ = xyz.shape
    idx = torch.empty(B, npoint, device=xyz.device, dtype=torch.int64)
    _lib.call("hitadv_fps_pct", _p(xyz), _p(start), B, N, npoint, _p(idx), _stream())
    return idx


# --------------------------------------------------------------------------- victim-side helper
def linear_max_bwd(dg, W, idx, N, act_out=None):
    """dX[b*N+n,:] = sum_{j: idx[b,j]==n} dg[b,j] * W[j,:]   (dg[B,Cout], W[Cout,Cin], idx[B,Cout] int64);
    with ``act_out`` (the ReLU'd forward maxima) only channels whose output is > 0 pass gradient."""
    dg, W = _dev(dg, "dg"), _dev(W, "W")
    idx = _dev(idx, "idx", torch.int64)
    B, Cout = dg.shape
    Cin = W.shape[1]
    dX = torch.empty(B * N, Cin, device=dg.device)
    _lib.call("hitadv_linear_max_bwd", _p(dg), _p(W), _p(idx), _p(act_out), B, N, Cout, Cin, _p(dX), _stream())
    return dX


def max_over_points(y, B, N, bias=None, relu=False):
    """y[B*N,C] (points-major) -> (act(max over the N points + bias) [B,C], arg-max [B,C] int64, lowest n on ties)."""
    y = _dev(y, "y")
    C = y.shape[-1]
    n = _lib.load().hitadv_max_over_points_scratch(B, C)
    pv = torch.empty(n, device=y.device)
    pi = torch.empty(n, device=y.device, dtype=torch.int32)
    out = torch.empty(B, C, device=y.device)
    idx = torch.empty(B, C, device=y.device, dtype=torch.int64)
    _lib.call("hitadv_max_over_points", _p(y), B, N, C, _p(bias), 1 if relu else 0, _p(pv), _p(pi), _p(out), _p(idx),
              _stream())
    return out, idx


def linear_max_fwd(x, Wt, B, N, bias=None, relu=False):
    """Fused x[B*N,Cin] @ Wt[Cin,Cout] -> act(max over the N points + bias) [B,Cout] and its arg-max [B,Cout] int64
    (f32 MFMA; the [B*N,Cout] activation is never written).  Cin in {64,128}, Cout % 64 == 0."""
    x, Wt = _dev(x, "x"), _dev(Wt, "Wt")
    Cin, Cout = Wt.shape
    n = _lib.load().hitadv_linear_max_fwd_scratch(B, N, Cout)
    pv = torch.empty(n, device=x.device)
    pi = torch.empty(n, device=x.device, dtype=torch.int32)
    out = torch.empty(B, Cout, device=x.device)
    idx = torch.empty(B, Cout, device=x.device, dtype=torch.int64)
    tickets = _fc_scratch_for(x, 1 << 14)  # zeroed, self-resetting; shared with fc_layer (never concurrent on a stream)
    _lib.call("hitadv_linear_max_fwd", _p(x), _p(Wt), _p(bias), B, N, Cin, Cout, 1 if relu else 0, _p(pv), _p(pi),
              _p(out), _p(idx), _p(tickets), _stream())
    return out, idx


def split_weights_bf16x3(Wr, out=None):
    """Wr [Cout,Cin] fp32 (row-major, one row per output channel) -> its three bf16 pieces [3,Cout,Cin] (int16 storage)
    that sum to it exactly; ``out`` = an existing buffer to refill (addresses held by a captured graph stay valid)."""
    Wr = _dev(Wr, "Wr")
    Cout, Cin = Wr.shape
    if out is None:
        out = torch.empty(3, Cout, Cin, device=Wr.device, dtype=torch.int16)
    _lib.call("hitadv_split_weights_bf16x3", _p(Wr), Cout, Cin, _p(out), _stream())
    return out


def linear_max_fwd_bf16x3(x, W3, B, N, bias=None, relu=False, blocks=0):
    """``linear_max_fwd`` on the bf16 matrix cores at fp32 accuracy (three-piece split of both operands, six cross terms
    in an fp32 accumulator; csrc/victim_bf3.hip).  W3 = split_weights_bf16x3(Wt.t()).  ``blocks``: workgroups the launch
    spreads over (0 = one per CU); results do not depend on it."""
    x = _dev(x, "x")
    if W3.device != x.device:
        raise RuntimeError("linear_max_fwd_bf16x3: weight pieces live on %s, the activations on %s" % (W3.device, x.device))
    _, Cout, Cin = W3.shape
    n = _lib.load().hitadv_linear_max_fwd_bf16x3_scratch(B, N, Cout, int(blocks))
    if n <= 0:
        raise RuntimeError("linear_max_fwd_bf16x3: bad sizes (B=%d, N=%d, Cout=%d, blocks=%d)" % (B, N, Cout, blocks))
    pv = torch.empty(n, device=x.device)
    pi = torch.empty(n, device=x.device, dtype=torch.int32)
    out = torch.empty(B, Cout, device=x.device)
    idx = torch.empty(B, Cout, device=x.device, dtype=torch.int64)
    tickets = _fc_scratch_for(x, 1 << 14)
    _lib.call("hitadv_linear_max_fwd_bf16x3", _p(x), _p(W3), _p(bias), B, N, Cin, Cout, 1 if relu else 0, int(blocks), _p(pv),
              _p(pi), _p(out), _p(idx), _p(tickets), _stream())
    return out, idx


def split_weights_f16x2(Wr, out=None, range_flag=None):
    """Wr [Cout,Cin] fp32 -> its two fp16 pieces [2,Cout,Cin] (int16 storage; the second scaled by 2^11), fragment order of
    ``split_weights_bf16x3``.  ``range_flag`` (int32[1] on the device) is raised by a weight beyond fp16's range."""
    Wr = _dev(Wr, "Wr")
    Cout, Cin = Wr.shape
    if out is None:
        out = torch.empty(2, Cout, Cin, device=Wr.device, dtype=torch.int16)
    _lib.call("hitadv_split_weights_f16x2", _p(Wr), Cout, Cin, _p(out), _p(range_flag), _stream())
    return out


def linear_max_fwd_f16x2(x, W2, B, N, bias=None, relu=False, blocks=0, range_flag=None, packed=False):
    """``linear_max_fwd`` on the fp16 matrix cores: two pieces per operand, three exact products per useful one, fp32
    accumulators (csrc/victim_bf3.hip, MODE 1) -- half the matrix time of the bf16x3 form, errors at fp32's own roundoff
    level.  W2 = split_weights_f16x2(Wt.t()).  ``range_flag``: int32[1] on the device, raised (never cleared) when an
    activation lies beyond fp16's range."""
    x = _dev(x, "x")
    if W2.device != x.device:
        raise RuntimeError("linear_max_fwd_f16x2: weight pieces live on %s, the activations on %s" % (W2.device, x.device))
    _, Cout, Cin = W2.shape
    n = _lib.load().hitadv_linear_max_fwd_bf16x3_scratch(B, N, Cout, int(blocks))
    if n <= 0:
        raise RuntimeError("linear_max_fwd_f16x2: bad sizes (B=%d, N=%d, Cout=%d, blocks=%d)" % (B, N, Cout, blocks))
    pv = torch.empty(n, device=x.device)
    pi = torch.empty(n, device=x.device, dtype=torch.int32)
    out = torch.empty(B, Cout, device=x.device)
    idx = torch.empty(B, Cout, device=x.device, dtype=torch.int64)
    tickets = _fc_scratch_for(x, 1 << 14)
    if packed:  # x holds packed pieces (one 32-bit word per value, written by pointnet_rowmlp_fwd(mode=2)): nothing to split
        _lib.call("hitadv_linear_max_fwd_f16x2_packed", _p(x), _p(W2), _p(bias), B, N, Cin, Cout, 1 if relu else 0, int(blocks),
                  _p(pv), _p(pi), _p(out), _p(idx), _p(tickets), _stream())
        return out, idx
    _lib.call("hitadv_linear_max_fwd_f16x2", _p(x), _p(W2), _p(bias), B, N, Cin, Cout, 1 if relu else 0, int(blocks), _p(pv),
              _p(pi), _p(out), _p(idx), _p(tickets), _p(range_flag), _stream())
    return out, idx


def linear_max_fwd_supported(Cin, Cout):
    return Cin in (64, 128) and Cout % 64 == 0


def fc_layer(x, Wt, bias=None, relu=False, mask=None):
    """act((x gated by mask > 0) @ Wt + bias): x [B,K], Wt [K,NOUT] -> [B,NOUT] (f32 MFMA, split-K over blocks,
    the last block of a tile to arrive reduces in chunk order: one launch, deterministic)."""
    x, Wt = _dev(x, "x"), _dev(Wt, "Wt")
    B, K = x.shape
    NOUT = Wt.shape[1]
    out = torch.empty(B, NOUT, device=x.device)
    scratch = _fc_scratch_for(x, _lib.load().hitadv_fc_layer_scratch_floats(B, K, NOUT))
    _lib.call("hitadv_fc_layer", _p(x), _p(mask), _p(Wt), _p(bias), B, K, NOUT, 1 if relu else 0, _p(out), _p(scratch),
              _stream())
    return out


def fc_layer_pre(pre, Wpre, Wt, bias=None, relu=False, mask=None):
    """``fc_layer`` whose input is evaluated on the way in: x = (sum_t pre[:, t, :]) @ Wpre  (pre [B,T,J], Wpre [J,K], J <= 64),
    gated by mask > 0, then @ Wt [K,NOUT] -- the sum over the tiles' partials, the first (tiny) layer of a backward stack and
    its second layer in one launch."""
    pre, Wpre, Wt = _dev(pre, "pre"), _dev(Wpre, "Wpre"), _dev(Wt, "Wt")
    B, T, J = pre.shape
    K, NOUT = Wt.shape
    if tuple(Wpre.shape) != (J, K):
        raise ValueError("Wpre must be [%d,%d]" % (J, K))
    out = torch.empty(B, NOUT, device=pre.device)
    scratch = _fc_scratch_for(pre, _lib.load().hitadv_fc_layer_scratch_floats(B, K, NOUT))
    _lib.call("hitadv_fc_layer_pre", _p(pre), T, J, _p(Wpre), _p(mask), _p(Wt), _p(bias), B, K, NOUT, 1 if relu else 0,
              _p(out), _p(scratch), _stream())
    return out


def sum_partials(part, extra=None):
    """part [B,T,M] (+ extra [B,M]) -> [B,M], summed in ascending T."""
    B, T, M = part.shape
    out = torch.empty(B, M, device=part.device)
    _lib.call("hitadv_sum_partials", _p(part), _p(extra), B, T, M, _p(out), _stream())
    return out


def pointnet_rowmlp_fwd(stage, B, N, W2, b2, o2, x=None, T=None, hin=None, W0=None, b0=None, W1=None, b1=None,
                        xp=None, o0=None, o1=None, mode=0, range_flag=None):
    """``mode`` 1: the layer products on the fp16 matrix cores, two pieces per operand; 2: the same with ``o2`` written as
    packed pieces for ``linear_max_fwd_f16x2(..., packed=True)`` (include/hitadv.h)."""
    _lib.call("hitadv_pointnet_rowmlp_fwd", stage, _p(x), _p(T), _p(hin), _p(W0), _p(b0), _p(W1), _p(b1), _p(W2),
              _p(b2), _p(xp), _p(o0), _p(o1), _p(o2), B, N, int(mode), _p(range_flag), _stream())


def pointnet_rowmlp_fwd_stn(B, N, x, F5, W6, b6, Tout, W0, b0, W1, b1, W2, b2, o0, o1, o2, xp=None, mode=0, range_flag=None):
    """Stage 1 of the forward chain with STN3d's last layer (F5 [B,256] @ W6 [256,9] + b6 -> Tout [B,9]) evaluated inside."""
    if F5.shape[1] != 256 or tuple(W6.shape) != (256, 9):
        raise ValueError("the fused input transform is the 256 -> 9 layer")
    _lib.call("hitadv_pointnet_rowmlp_fwd_stn", _p(x), _p(F5), _p(W6), _p(b6), _p(Tout), _p(W0), _p(b0), _p(W1), _p(b1),
              _p(W2), _p(b2), _p(xp), _p(o0), _p(o1), _p(o2), B, N, int(mode), _p(range_flag), _stream())


def pointnet_rowmlp_fwd_deform(B, N, ori, central, perturb, sigma, adv, inv_den, W0, b0, W2, b2, o0, o2, mode=0, range_flag=None):
    """Stage 0 of the forward chain on HiT-ADV's deformation of ``ori``, evaluated inside (``adv`` and ``inv_den`` are
    OUTPUTS: what ``deform_fwd_into`` writes)."""
    C = sigma.shape[1]
    if C > 256:
        raise ValueError("the fused deformation holds at most 256 centres")
    _lib.call("hitadv_pointnet_rowmlp_fwd_deform", _p(ori), _p(central), _p(perturb), _p(sigma), C, _p(adv), _p(inv_den),
              _p(W0), _p(b0), _p(W2), _p(b2), _p(o0), _p(o2), B, N, int(mode), _p(range_flag), _stream())


def pointnet_rowmlp_bwd(stage, B, N, dg, idx, W3r, A2, W2r, out, gmask=None, A1=None, W1r=None, H1=None, dH1in=None,
                        W0r=None, T=None, x=None, dPin=None, dTpart=None, pres_in=None, pres_out=None, mode=0, overflow=None,
                        words=1, dTfix=None):
    """``pres_in`` / ``pres_out``: int64 [B, tiles, words] row-presence bit sets handed from stage to stage (see hitadv.h,
    ``pointnet_rowmlp_bwd_tiles``);
    ``mode`` 1: the products on the fp16 matrix cores, two pieces per operand; ``words``: 64-point words per block (the tables'
    last dimension); ``overflow``: int32 [B, tiles] scratch, needed with two words."""
    if overflow is None and words > 1:
        overflow = torch.empty(B, pointnet_rowmlp_bwd_tiles(B, N, mode, words)[0], device=dg.device, dtype=torch.int32)
    if words > 1 and dTfix is None and stage > 0 and dTpart is not None:
        # scratch of the second launch (include/hitadv.h: hitadv_pointnet_rowmlp_bwd_fix): two one-word partials per tile
        dTfix = torch.empty(B, 2 * dTpart.shape[1], dTpart.shape[2], device=dg.device)
    _lib.call("hitadv_pointnet_rowmlp_bwd_fix", stage, _p(dg), _p(gmask), _p(idx), _p(W3r), W3r.shape[0], _p(A2), _p(W2r),
              _p(A1), _p(W1r), _p(H1), _p(dH1in), _p(W0r), _p(T), _p(x), _p(dPin), _p(dTpart), _p(out), _p(pres_in),
              _p(pres_out), _p(overflow), int(words), B, N, int(mode), _p(dTfix), _stream())


def pointnet_rowmlp_tiles(N):
    return int(_lib.load().hitadv_pointnet_rowmlp_tiles(N))


def pointnet_rowmlp_form(form=-1):
    """0: the streaming forward kernel, 1: one 64-point tile per workgroup, 2 (default): streaming except for stage 0 with the
    deformation inside (the same bits in every form; include/hitadv.h).  Process-wide; returns the previous value; no argument
    only reads."""
    return int(_lib.load().hitadv_pointnet_rowmlp_form(int(form)))


def pointnet_rowmlp_bwd_tiles(B, N, mode=0, words=None):
    """(tiles, words): a block of ``pointnet_rowmlp_bwd`` covers ``words`` 64-point words of a cloud (None: what the library
    recommends for this launch size); the per-tile partials are [B, tiles, .] and the row-presence tables int64
    [B, tiles, words] (include/hitadv.h)."""
    lib = _lib.load()
    if words is None:
        words = int(lib.hitadv_pointnet_rowmlp_bwd_words(int(B), int(N), int(mode)))
    return int(lib.hitadv_pointnet_rowmlp_bwd_tiles(int(N), int(words))), int(words)


class EdgeMax(torch.autograd.Function):
    """out[b,i,:] = lrelu(V[b,i,:] + max_{j in idx[b,i,:]} U[b,j,:])  -- EdgeConv's gather / max / activation after the
    1x1 convolution has been split into two per-point products (see hitadv_edge_max_fwd).  UV [B,N,2C] = [U | V] (the
    two products come out of one GEMM), idx [B,N,k]; the gradient comes back as [dU | dV] for one GEMM too."""

    @staticmethod
    def forward(ctx, UV, idx, slope):
        UV = _dev(UV, "UV")
        idx = _dev(idx, "idx", torch.int64)
        B, N, C2 = UV.shape
        C = C2 // 2
        out = torch.empty(B, N, C, device=UV.device)
        arg = torch.empty(B, N, C, device=UV.device, dtype=torch.int32)
        _lib.call("hitadv_edge_max_fwd", _p(UV), ctypes.c_void_p(UV.data_ptr() + 4 * C), C2, _p(idx), B, N, C,
                  idx.shape[2], ctypes.c_float(slope), _p(out), _p(arg), _stream())
        ctx.save_for_backward(out, arg, idx)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, dout):
        out, arg, idx = ctx.saved_tensors
        B, N, C = out.shape
        k = idx.shape[2]
        dout = dout.contiguous()
        dUV = torch.empty(B, N, 2 * C, device=out.device)
        scratch = torch.empty(B * (2 * N + N * k), device=out.device, dtype=torch.int32)
        _lib.call("hitadv_edge_max_bwd", _p(dout), _p(out), _p(arg), _p(idx), B, N, C, k, ctypes.c_float(ctx.slope),
                  _p(dUV), ctypes.c_void_p(dUV.data_ptr() + 4 * C), 2 * C, _p(scratch), _stream())
        return dUV, None, None


def edge_max(U, V, idx, slope=0.2):
    """U, V [B,N,C] separate tensors (concatenated here); ``edge_max_fused`` takes the [U | V] product directly."""
    return EdgeMax.apply(torch.cat((U, V), dim=2), idx, slope)


def edge_max_fused(UV, idx, slope=0.2):
    return EdgeMax.apply(UV, idx, slope)


class GroupAddReLU(torch.autograd.Function):
    """H[b,i,s,:] = relu(U[b, idx[b,i,s], :] + V[b,i,:]) -- the first shared layer of a sample-and-group block after the
    1x1 convolution has been split over the neighbour and the centre (hitadv_group_add_relu_fwd).  U [B,N,C], V [B,S,C],
    idx [B,S,ns] -> [B,S,ns,C]."""

    @staticmethod
    def forward(ctx, U, V, idx):
        U, V = _dev(U, "U"), _dev(V, "V")
        idx = _dev(idx, "idx", torch.int64)
        B, N, C = U.shape
        S, ns = idx.shape[1], idx.shape[2]
        H = torch.empty(B, S, ns, C, device=U.device)
        _lib.call("hitadv_group_add_relu_fwd", _p(U), _p(V), _p(idx), B, N, S, ns, C, _p(H), _stream())
        ctx.save_for_backward(U, V, idx)
        return H

    @staticmethod
    def backward(ctx, dH):
        U, V, idx = ctx.saved_tensors
        B, N, C = U.shape
        S, ns = idx.shape[1], idx.shape[2]
        dH = dH.contiguous()
        dU, dV = torch.empty_like(U), torch.empty_like(V)
        scratch = torch.empty(B * (2 * N + S * ns), device=U.device, dtype=torch.int32)
        _lib.call("hitadv_group_add_relu_bwd", _p(dH), _p(U), _p(V), _p(idx), B, N, S, ns, C, _p(dU), _p(dV), _p(scratch),
                  _stream())
        return dU, dV, None


def group_add_relu(U, V, idx):
    return GroupAddReLU.apply(U, V, idx)


def group_add_relu_linear_supported(C, Cout, S, ns):
    return bool(_lib.load().hitadv_group_add_relu_linear_supported(int(C), int(Cout), int(S), int(ns)))


class GroupAddReLULinear(torch.autograd.Function):
    """relu(relu(U[b, idx] + V[b,i]) W^T + bias) over the grouped rows -> [B,S,ns,Cout]: ``group_add_relu`` and the shared layer behind
    it as one kernel (hitadv_group_add_relu_linear; the [B,S,ns,C] activation between them never exists).  The ReLU backward of the
    output is left to the consumer (the gradient arrives gated, as for ``_LinearReLUGatedLater``); the backward pass is the input
    gradient of the layer (hitadv_rows_linear on the transposed pieces) followed by ``group_add_relu``'s own."""

    @staticmethod
    def forward(ctx, U, V, idx, W2, Wt2, bias, flag):
        U, V = _dev(U, "U"), _dev(V, "V")
        idx = _dev(idx, "idx", torch.int64)
        B, N, C = U.shape
        S, ns = idx.shape[1], idx.shape[2]
        Cout = W2.shape[1]
        Y = torch.empty(B, S, ns, Cout, device=U.device)
        _lib.call("hitadv_group_add_relu_linear", _p(U), _p(V), _p(idx), B, N, S, ns, C, _p(W2), _p(bias), Cout, 1, _p(Y), _p(flag),
                  _stream())
        ctx.save_for_backward(U, V, idx, Wt2)
        ctx.flag = flag
        return Y

    @staticmethod
    def backward(ctx, g):
        U, V, idx, Wt2 = ctx.saved_tensors
        B, N, C = U.shape
        S, ns = idx.shape[1], idx.shape[2]
        dH = rows_linear(g.reshape(-1, g.shape[-1]).contiguous(), Wt2, None, False, ctx.flag)
        dU, dV = torch.empty_like(U), torch.empty_like(V)
        scratch = torch.empty(B * (2 * N + S * ns), device=U.device, dtype=torch.int32)
        _lib.call("hitadv_group_add_relu_bwd", _p(dH), _p(U), _p(V), _p(idx), B, N, S, ns, C, _p(dU), _p(dV), _p(scratch), _stream())
        return dU, dV, None, None, None, None, None


def group_add_relu_supported(C, ns):
    return C % 4 == 0 and ns <= 64


class LReluPool(torch.autograd.Function):
    """[max_i lrelu(Z[b,i,:]) | mean_i lrelu(Z[b,i,:])] -> [B,2C] from the pre-activation Z [B,N,C] (hitadv_lrelu_pool_fwd)."""

    @staticmethod
    def forward(ctx, Z, slope):
        Z = _dev(Z, "Z")
        B, N, C = Z.shape
        out = torch.empty(B, 2 * C, device=Z.device)
        arg = torch.empty(B, C, device=Z.device, dtype=torch.int32)
        _lib.call("hitadv_lrelu_pool_fwd", _p(Z), B, N, C, ctypes.c_float(slope), _p(out), _p(arg), _stream())
        ctx.save_for_backward(Z, arg)
        ctx.slope = slope
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g, _garg):
        Z, arg = ctx.saved_tensors
        B, N, C = Z.shape
        g = g.contiguous()
        dZ = torch.empty_like(Z)
        _lib.call("hitadv_lrelu_pool_bwd", _p(Z), _p(g), _p(arg), B, N, C, ctypes.c_float(ctx.slope), _p(dZ), _stream())
        return dZ, None


def group_linear_max_supported(Cin, Cout, ns):
    return bool(_lib.load().hitadv_group_linear_max_supported(int(Cin), int(Cout), int(ns)))


class GroupLinearMax(torch.autograd.Function):
    """out[g,c] = relu(max_j x[g,j,:] . W[c,:] + b[c]) for x [G, ns, Cin] (the last shared layer of a sample-and-group block
    and the max over the neighbours, model/pointnet2_utils.py:197-201) without the [G*ns, Cout] activation: fp16x2 MFMA
    forward with the max / arg-max in its epilogue, and a backward that scatters the G*Cout routed gradient values into a
    zero A operand in LDS and multiplies by W on the matrix cores (csrc/group_mlp.hip).  ``Wr`` [Cout,Cin], ``bias`` [Cout]
    are constants (the folded eval-mode layer); the gradient goes to ``x`` only."""

    @staticmethod
    def forward(ctx, x, Wr, bias, range_flag, W2=None, Wb2=None, relu_input=False):
        x = _dev(x, "x")
        G, ns, Cin = x.shape
        Cout = Wr.shape[0]
        if W2 is None:
            W2 = split_weights_f16x2(Wr.contiguous(), range_flag=range_flag)            # forward operand
            Wb2 = split_weights_f16x2(Wr.t().contiguous(), range_flag=range_flag)        # backward operand: pieces of Wt [Cin,Cout]
        out = torch.empty(G, Cout, device=x.device)
        arg = torch.empty(G, Cout, device=x.device, dtype=torch.int32)
        _lib.call("hitadv_group_linear_max_fwd", _p(x), _p(W2), _p(bias), G, ns, Cin, Cout, _p(out), _p(arg), _p(range_flag),
                  _stream())
        # ``relu_input``: x is the ReLU output of the layer in front and the caller wants THAT layer's ReLU backward applied to
        # the gradient this node returns (dX gated by x > 0 on its way out of the kernel; the producer must then not gate again)
        ctx.relu_input = bool(relu_input)
        ctx.save_for_backward(out, arg, Wb2, *((x,) if relu_input else ()))
        ctx.dims = (G, ns, Cin, Cout)
        ctx.range_flag = range_flag
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g, _ga):
        out, arg, Wb2 = ctx.saved_tensors[:3]
        G, ns, Cin, Cout = ctx.dims
        dX = torch.empty(G, ns, Cin, device=out.device)
        if ctx.relu_input:
            _lib.call("hitadv_group_linear_max_bwd_masked", _p(g.contiguous()), _p(out), _p(arg), _p(Wb2), G, ns, Cin, Cout,
                      _p(ctx.saved_tensors[3]), _p(dX), _p(ctx.range_flag), _stream())
        else:
            _lib.call("hitadv_group_linear_max_bwd", _p(g.contiguous()), _p(out), _p(arg), _p(Wb2), G, ns, Cin, Cout, _p(dX),
                      _p(ctx.range_flag), _stream())
        return dX, None, None, None, None, None, None


def group_linear_max_g16_supported(Cin, Cout, ns):
    """The same fused layer on the tiled GEMM core (csrc/gemm16.hip) -- the widths the register-resident kernels do not cover."""
    return bool(_lib.load().hitadv_group_linear_max_g16_supported(int(Cin), int(Cout), int(ns)))


class GroupLinearMaxG16(torch.autograd.Function):
    """``GroupLinearMax`` on ``gemm_f16x2_k``: Wp = split_rows_f16x2(Wr [Cout,Cin]), Wtp = split_rows_f16x2(Wr.t())."""

    @staticmethod
    def forward(ctx, x, Wp, Wtp, bias, range_flag, relu_input=False):
        x = _dev(x, "x")
        G, ns, Cin = x.shape
        Cout = Wp.shape[1]
        out = torch.empty(G, Cout, device=x.device)
        arg = torch.empty(G, Cout, device=x.device, dtype=torch.int32)
        _lib.call("hitadv_group_linear_max_g16_fwd", _p(x), _p(Wp), _p(bias), G, ns, Cin, Cout, _p(out), _p(arg), _p(range_flag),
                  _stream())
        ctx.relu_input = bool(relu_input)
        ctx.save_for_backward(out, arg, Wtp, *((x,) if relu_input else ()))
        ctx.dims, ctx.range_flag = (G, ns, Cin, Cout), range_flag
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g, _ga):
        out, arg, Wtp = ctx.saved_tensors[:3]
        G, ns, Cin, Cout = ctx.dims
        dm = torch.where(out > 0, g, torch.zeros_like(g)).contiguous()  # [G,Cout]: the layer's own ReLU
        dX = torch.empty(G, ns, Cin, device=out.device)
        _lib.call("hitadv_group_linear_max_g16_bwd", _p(dm), _p(arg), _p(Wtp), G, ns, Cin, Cout,
                  _p(ctx.saved_tensors[3]) if ctx.relu_input else None, _p(dX), _p(ctx.range_flag), _stream())
        return dX, None, None, None, None, None


def group_linear_max_g16(x, Wp, Wtp, bias, range_flag=None, return_arg=False, relu_input=False):
    lead = x.shape[:-2]
    out, arg = GroupLinearMaxG16.apply(x.reshape(-1, x.shape[-2], x.shape[-1]), Wp, Wtp, bias, range_flag, relu_input)
    out = out.view(*lead, out.shape[-1])
    return (out, arg.view(*lead, arg.shape[-1])) if return_arg else out


def group_linear_max(x, Wr, bias, range_flag=None, return_arg=False, pieces=None, relu_input=False):
    """x [..., ns, Cin] -> relu(max over the ns rows of (x W^T + bias)) [..., Cout]; see ``GroupLinearMax``."""
    lead = x.shape[:-2]
    W2, Wb2 = pieces if pieces is not None else (None, None)  # ``pieces``: (split_weights_f16x2(Wr), split_weights_f16x2(Wr.t())) kept by the caller
    out, arg = GroupLinearMax.apply(x.reshape(-1, x.shape[-2], x.shape[-1]), Wr, bias, range_flag, W2, Wb2, relu_input)
    out = out.view(*lead, out.shape[-1])
    return (out, arg.view(*lead, arg.shape[-1])) if return_arg else out


class _PointsMajor(torch.autograd.Function):
    """[B,C,N] -> contiguous [B,N,C] for a few channels (hitadv_transpose_small); the backward pass is the same kernel the other way."""

    @staticmethod
    def forward(ctx, x, to_points_major):
        x = _dev(x, "x")
        B, P, Q = x.shape
        C, N = (P, Q) if to_points_major else (Q, P)
        y = torch.empty(B, Q, P, device=x.device)
        _lib.call("hitadv_transpose_small", _p(x), _p(y), B, C, N, 1 if to_points_major else 0, _stream())
        ctx.dir = to_points_major
        return y

    @staticmethod
    def backward(ctx, g):
        return _PointsMajor.apply(g.contiguous(), not ctx.dir), None


def points_major(x):
    """``x.permute(0, 2, 1).contiguous()`` for x [B,C,N] with a handful of channels (coordinates); differentiable.  CPU tensors and
    wide inputs take torch's own path."""
    if x.is_cuda and x.dim() == 3 and x.shape[1] <= 16 and x.dtype == torch.float32:
        return _PointsMajor.apply(x.contiguous(), True)
    return x.permute(0, 2, 1).contiguous()


def rows_linear_supported(Cin, Cout):
    return bool(_lib.load().hitadv_rows_linear_supported(int(Cin), int(Cout)))


def rows_linear(x2, W2, bias=None, relu=False, range_flag=None):
    """act(x2 W^T + bias) for x2 [rows, Cin] (very many rows, Cin / Cout in {64, 128}) on the fp16 matrix cores at fp32 accuracy
    (hitadv_rows_linear).  W2 = split_weights_f16x2(W [Cout, Cin]); no autograd (model/_pointwise.py wraps it)."""
    x2 = _dev(x2, "x2")
    rows, Cin = x2.shape
    Cout = W2.shape[1]
    y = torch.empty(rows, Cout, device=x2.device)
    _lib.call("hitadv_rows_linear", _p(x2), _p(W2), _p(bias), rows, Cin, Cout, 1 if relu else 0, _p(y), _p(range_flag), _stream())
    return y


def bmm_supported(M, N, K):
    return bool(_lib.load().hitadv_bmm_f32_supported(int(M), int(N), int(K)))


def _bmm_raw(a, b, ta, tb):
    Bn = a.shape[0]
    M, K = (a.shape[2], a.shape[1]) if ta else (a.shape[1], a.shape[2])
    N = b.shape[1] if tb else b.shape[2]
    c = torch.empty(Bn, M, N, device=a.device)
    _lib.call("hitadv_bmm_f32", _p(a), _p(b), _p(c), Bn, M, N, K, 1 if ta else 0, 1 if tb else 0, _stream())
    return c


class Bmm(torch.autograd.Function):
    """C[b] = op(a[b]) op(b[b]) for small batched fp32 matrices (hitadv_bmm_f32: exact fp32 chains, 64 x 64 tiles); ``ta`` / ``tb``:
    the operand is STORED transposed ([K,M] / [N,K]).  The backward pass is four more calls of the same kernel."""

    @staticmethod
    def forward(ctx, a, b, ta, tb):
        a, b = _dev(a, "a").contiguous(), _dev(b, "b").contiguous()
        ctx.save_for_backward(a, b)
        ctx.t = (bool(ta), bool(tb))
        return _bmm_raw(a, b, ta, tb)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ta, tb = ctx.t
        g = g.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _bmm_raw(b, g, tb, True) if ta else _bmm_raw(g, b, False, not tb)
        if ctx.needs_input_grad[1]:
            db = _bmm_raw(g, a, True, ta) if tb else _bmm_raw(a, g, not ta, False)
        return da, db, None, None


def bmm(a, b, trans_a=False, trans_b=False):
    """``torch.bmm(op(a), op(b))`` where every dimension allows the tiled kernel (multiples of 64 / 64 / 32), else torch's."""
    M, K = (a.shape[2], a.shape[1]) if trans_a else (a.shape[1], a.shape[2])
    N = b.shape[1] if trans_b else b.shape[2]
    if a.is_cuda and a.dtype == torch.float32 and min(M, N, K) >= 64 and M % 64 == 0 and N % 64 == 0 and K % 64 == 0:
        return Bmm.apply(a, b, trans_a, trans_b)  # K % 64: the backward products contract over M or N and tile over K
    return torch.bmm(a.transpose(1, 2) if trans_a else a, b.transpose(1, 2) if trans_b else b)


class OffsetAttentionNorm(torch.autograd.Function):
    """PCT's offset attention between its two batched products (model/pct_cls.py:127-131): ``softmax(E, -1)`` followed by the
    column renormalisation ``A / (1e-9 + A.sum(dim=1, keepdim=True))`` -- two launches forward and two backward instead of
    torch's four and nine element-wise passes (csrc/attention.hip)."""

    @staticmethod
    def forward(ctx, E):
        E = _dev(E, "E").contiguous()
        B, N, _ = E.shape
        A = torch.empty_like(E)
        c = torch.empty(B, N, device=E.device)
        _lib.call("hitadv_offset_attention_fwd", _p(E), B, N, _p(A), _p(c), _stream())
        ctx.save_for_backward(A, c)
        return A

    @staticmethod
    def backward(ctx, dA):
        A, c = ctx.saved_tensors
        B, N, _ = A.shape
        dA = dA.contiguous()
        dE, h = torch.empty_like(A), torch.empty_like(c)
        _lib.call("hitadv_offset_attention_bwd", _p(dA), _p(A), _p(c), B, N, _p(h), _p(dE), _stream())
        return dE


def offset_attention_norm(E):
    """``A = softmax(E, -1); A / (1e-9 + A.sum(1, keepdim=True))`` for E [B,N,N] (fused where N allows, else torch's ops)."""
    if (E.is_cuda and E.dtype == torch.float32 and E.dim() == 3 and E.shape[1] == E.shape[2]
            and bool(_lib.load().hitadv_offset_attention_supported(int(E.shape[2])))):
        return OffsetAttentionNorm.apply(E)
    A = torch.softmax(E, dim=-1)
    return A / (1e-9 + A.sum(dim=1, keepdim=True))


class OffsetAttentionLayer(torch.autograd.Function):
    """One offset-attention layer of PCT (model/pct_cls.py:111-139) on points-major x [B,N,C] with its constants folded:
    ``q = x Wq^T`` (q and k share their weight), ``A = offset_attention_norm(q q^T)``, ``v = x Wv^T + bv``, ``x_r = A^T v``,
    ``out = x + relu((x - x_r) Wt^T + bt)`` (trans_conv with its BatchNorm folded).  The forward pass is the composition
    ``SA_Layer.forward_pm`` used to write out; the BACKWARD pass is written by hand: autograd's own spent ten element-wise launches
    per layer on glue (the ReLU mask, ``-dt``, four accumulations of the gradient of ``x``, the sum of the two gradients of ``q``):
    here the signs ride on ``addmm(alpha=-1)``, the residual's gradient is the first ``addmm``'s addend, and what is left is one mask,
    one add for ``dq`` and one for ``dt``.  The weights are constants (gradient with respect to ``x`` only)."""

    @staticmethod
    def forward(ctx, x, Wq, Wv, bv, Wt, bt):
        x = _dev(x, "x").contiguous()
        B, N, C = x.shape
        x2 = x.view(B * N, C)
        q = x2 @ Wq.t()
        q3 = q.view(B, N, -1)
        E = _bmm_raw(q3, q3, False, True)
        A = torch.empty_like(E)
        c = torch.empty(B, N, device=x.device)
        _lib.call("hitadv_offset_attention_fwd", _p(E), B, N, _p(A), _p(c), _stream())
        v = torch.addmm(bv, x2, Wv.t())
        x_r = _bmm_raw(A, v.view(B, N, C), True, False)
        y = torch._addmm_activation(bt, (x - x_r).view(B * N, C), Wt.t(), use_gelu=False)
        ctx.save_for_backward(q, A, c, v, y, Wq, Wv, Wt)
        return x + y.view(B, N, C)

    @staticmethod
    def backward(ctx, g):
        q, A, c, v, y, Wq, Wv, Wt = ctx.saved_tensors
        B, N, _ = A.shape
        C = v.shape[1]
        g2 = g.contiguous().view(B * N, C)
        dt = torch.ops.aten.threshold_backward(g2, y, 0) @ Wt            # d (x - x_r): the gradient of x_r is -dt
        dt3, q3 = dt.view(B, N, C), q.view(B, N, -1)
        dv_neg = _bmm_raw(A, dt3, False, False)                            # -(d v)  = A dt
        dA_neg = _bmm_raw(v.view(B, N, C), dt3, False, True)               # -(d A)  = v dt^T
        dE_neg, h = torch.empty_like(A), torch.empty_like(c)               # the normalisation's backward is linear: -(d E)
        _lib.call("hitadv_offset_attention_bwd", _p(dA_neg), _p(A), _p(c), B, N, _p(h), _p(dE_neg), _stream())
        dq_neg = _bmm_raw(dE_neg, q3, False, False) + _bmm_raw(dE_neg, q3, True, False)   # -(dE + dE^T) q
        dx = torch.addmm(g2, dv_neg.view(B * N, C), Wv, alpha=-1.0)        # g (the residual) + dv Wv
        dx = torch.addmm(dx, dq_neg.view(B * N, -1), Wq, alpha=-1.0)       # + dq Wq
        dx += dt
        return dx.view(B, N, C), None, None, None, None, None


def offset_attention_layer_supported(N, C, Cq):
    lib = _lib.load()
    return bool(N % 64 == 0 and C % 64 == 0 and Cq % 64 == 0 and lib.hitadv_offset_attention_supported(int(N)))


def offset_attention_layer(x, Wq, Wv, bv, Wt, bt):
    return OffsetAttentionLayer.apply(x, Wq, Wv, bv, Wt, bt)


def gemm_f16x2_supported(N, K):
    """Whether ``gemm_f16x2`` / ``linear_lrelu_pool`` are built for N output columns over a K-deep contraction."""
    return bool(_lib.load().hitadv_gemm_f16x2_supported(int(N), int(K)))


def split_rows_f16x2(W, range_flag=None):
    """W [N,K] fp32 (N output columns) -> the two fp16 pieces [2,N,K] (int16 storage) the fp16x2 GEMMs take as their B operand."""
    W = _dev(W.detach(), "W")
    N, K = W.shape
    Wp = torch.empty(2, N, K, device=W.device, dtype=torch.int16)
    _lib.call("hitadv_split_rows_f16x2", _p(W), N, K, _p(Wp), _p(range_flag), _stream())
    return Wp


def gemm_f16x2(x, Wp, bias=None, relu=False, mask=None, range_flag=None):
    """act((x . [mask > 0]) W^T + bias) for x [M,K], Wp = split_rows_f16x2(W [N,K]): an fp32-accurate GEMM on the fp16 matrix
    cores (two pieces per operand, three exact products; include/hitadv.h).  No autograd: the callers write their chain rule."""
    x = _dev(x, "x")
    M, K = x.shape
    _, N, K2 = Wp.shape
    if K2 != K or not gemm_f16x2_supported(N, K):
        raise ValueError("gemm_f16x2: x [%d,%d] against pieces [2,%d,%d]" % (M, K, N, K2))
    if mask is not None:
        mask = _dev(mask, "mask")
    out = torch.empty(M, N, device=x.device)
    _lib.call("hitadv_gemm_f16x2", _p(x), _p(mask), _p(Wp), _p(bias), M, N, K, 1 if relu else 0, _p(out), _p(range_flag), _stream())
    return out


class LinearLReluPool(torch.autograd.Function):
    """DGCNN's embedding layer fused with its activation and poolings: x [B*npts,Cin] -> [max_p | mean_p] of
    lrelu(x W^T + bias) [B,2C].  The [B*npts,C] activation never exists; the backward pass rebuilds the gradient in front of
    the layer from a one-bit-per-value sign mask and the arg-max table inside the input-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, Wp, Wtp, bias, B, npts, slope, range_flag):
        x = _dev(x, "x")
        _, C, Cin = Wp.shape
        n = int(_lib.load().hitadv_linear_lrelu_pool_scratch(B, npts, C))
        pmax, psum = torch.empty(n, device=x.device), torch.empty(n, device=x.device)
        parg = torch.empty(n, device=x.device, dtype=torch.int32)
        bits = torch.empty(B * npts, C // 32, device=x.device, dtype=torch.int32)
        out = torch.empty(B, 2 * C, device=x.device)
        arg = torch.empty(B, C, device=x.device, dtype=torch.int32)
        _lib.call("hitadv_linear_lrelu_pool_fwd", _p(x), _p(Wp), _p(bias), B, npts, Cin, C, ctypes.c_float(slope), _p(pmax),
                  _p(psum), _p(parg), _p(bits), _p(out), _p(arg), _p(range_flag), _stream())
        ctx.save_for_backward(arg, bits, Wtp)
        ctx.dims, ctx.slope, ctx.flag = (B, npts, Cin, C), slope, range_flag
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g, _):
        arg, bits, Wtp = ctx.saved_tensors
        B, npts, Cin, C = ctx.dims
        dX = torch.empty(B * npts, Cin, device=g.device)
        _lib.call("hitadv_linear_lrelu_pool_bwd", _p(g.contiguous()), _p(arg), _p(bits), _p(Wtp), B, npts, Cin, C,
                  ctypes.c_float(ctx.slope), _p(dX), _p(ctx.flag), _stream())
        return dX, None, None, None, None, None, None, None


def linear_lrelu_pool(x, Wp, Wtp, bias, B, npts, slope=0.2, range_flag=None, return_arg=False):
    """x [B*npts,Cin]; Wp = split_rows_f16x2(W [C,Cin]), Wtp = split_rows_f16x2(W^T [Cin,C]) -> out [B,2C] (max | mean)."""
    out, arg = LinearLReluPool.apply(x.contiguous(), Wp, Wtp, bias, B, npts, slope, range_flag)
    return (out, arg) if return_arg else out


def lrelu_pool(Z, slope=0.2, return_arg=False):
    """``return_arg``: also the int32 [B,C] table of the points the maxima were taken at (the kernel's own tie rule)."""
    out, arg = LReluPool.apply(Z, slope)
    return (out, arg) if return_arg else out


def lrelu_pool_supported(C):
    return C % 64 == 0


def knn_features(x, K):
    """x [B,N,D] points-major features (D in {64,128}) -> idx [B,N,K] int64 of the K nearest points in feature space
    (closest first, ties -> lower index); no [B,N,N] score matrix (hitadv_knn_features)."""
    x = _dev(x.detach(), "x")
    B, N, D = x.shape
    xx = torch.empty(B, N, device=x.device)
    _lib.call("hitadv_row_sqnorm", _p(x), B * N, D, _p(xx), _stream())
    idx = torch.empty(B, N, K, device=x.device, dtype=torch.int64)
    _lib.call("hitadv_knn_features", _p(x), _p(xx), B, N, D, K, _p(idx), _stream())
    return idx


def knn_features_supported(D, K):
    return D in (64, 128) and K <= 20


def topk_rows(P, K, largest=True):
    """Row-wise top-K of a matrix [..., M] -> (vals[..., K], idx[..., K] int64), sorted, ties -> lower column."""
    P = _dev(P.detach(), "P")
    M = P.shape[-1]
    rows = P.numel() // M
    vals = torch.empty(*P.shape[:-1], K, device=P.device)
    idx = torch.empty(*P.shape[:-1], K, device=P.device, dtype=torch.int64)
    _lib.call("hitadv_topk_rows", _p(P), rows, M, K, 1 if largest else 0, _p(vals), _p(idx), _stream())
    return vals, idx


def radius_squared(radius):
    """The threshold of ``sqrdists > radius ** 2`` (model/pointnet2_utils.py:102): torch compares an fp32 tensor with a
    Python double in fp32, i.e. with the fp32 value of the DOUBLE square -- not with the fp32 product."""
    return struct.unpack('f', struct.pack('f', float(radius) ** 2))[0]


def query_ball_point(radius, nsample, xyz, new_xyz, reference=None):
    """torch-semantics ball query (model/pointnet2_utils.py:87-107): xyz[B,N,3], new_xyz[B,S,3] -> idx[B,S,nsample] int64.
    ``reference`` (None: the ``victim_reference_arithmetic`` switch, on by default) evaluates the distances as the
    reference's Gram-form ``square_distance`` does, bit for bit: the table is then the reference's."""
    xyz, new_xyz = _dev(xyz.detach(), "xyz"), _dev(new_xyz.detach(), "new_xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty(B, S, nsample, device=xyz.device, dtype=torch.int64)
    form = FORM_SQUARE_DISTANCE if _victim_form(reference) else FORM_DIRECT
    _lib.call("hitadv_query_ball_point_victim", B, N, S, ctypes.c_float(radius_squared(radius)), nsample, form,
              _p(new_xyz), _p(xyz), _p(idx), _stream())
    return idx
