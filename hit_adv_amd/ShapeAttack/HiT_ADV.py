"""HiT-ADV attack, MI355X-native.

Same public surface as the reference's ShapeAttack/HiT_ADV.py::HiT_ADV (ctor :18-42,
``attack(data[B,N,6], target[B]) -> (float64 ndarray [B,N,3], 0-d int64 tensor)`` :44-287),
re-designed around the hardware instead of translated:

* the kernel-density matrix and the C-step deformation loop (:160-175, :298-304) are ONE HIP
  kernel forward and one backward (ops.Deform) -- no [B,C,N] tensor, no 192-node autograd tape;
* the per-iteration device->host copies and the Python best-tracking loop (:186-217) are ONE HIP
  kernel (ops.best_update) working on device-resident buffers;
* the two-group Adam of :142-145 is ONE HIP kernel (ops.adam_step) with a device-side step counter;
* the bisection of the distance weight (:264-273) is a handful of [B]-sized device ops;
* therefore an inner iteration has no host synchronisation and is captured ONCE into a hipGraph
  (torch.cuda.CUDAGraph) that is replayed binary_step x num_iter times; the victim's
  forward/backward stay ordinary PyTorch-ROCm ops inside that graph.

Random draws come from the global CPU generator in the reference's order (randint for the FPS
start :501, then per binary step rand(B,C,3) :130 and rand(B,C) :133), so a seeded run follows the
reference's trajectory.

Deliberate deviations (none changes a returned value): victim parameter ``.grad`` fields are not
populated (only d loss / d (perturb, sigma) is computed); ``num_iter < 5`` does not raise
ZeroDivisionError (quirk Q4); the per-100-iteration stopwatch lines are not printed.
"""
import warnings

import os

import torch
import torch.nn.functional as F

from .. import ops
from ..model._pointwise import degrade_on_fp16_range
from ..model import _sampling
from ..pytorch3d_ops import knn_gather, knn_points
from ..util.dist_utils import ChamferDist, curvature_std


def _take(points, idx):
    """points[B,N,C], idx[B,S] -> [B,S,C]."""
    return points.gather(1, idx.unsqueeze(-1).expand(-1, -1, points.shape[2]))


def _buffers(G, B, N, C, dev):
    """Every device buffer of one attack, with a leading group dimension G (1 for a plain workspace): name -> tensor."""
    f = dict(device=dev, dtype=torch.float32)
    i64 = dict(device=dev, dtype=torch.int64)
    return dict(
        ori=torch.empty(G, B, 3, N, **f), central=torch.empty(G, B, 3, C, **f),
        hide_ref=torch.empty(G, B, C, **f),  # min-max normalised central kappa-std (constant)
        target=torch.empty(G, B, **i64),
        P=torch.zeros(G, B, C, 3, **f), sigma=torch.ones(G, B, C, **f),
        m_p=torch.zeros(G, B, C, 3, **f), v_p=torch.zeros(G, B, C, 3, **f),
        m_s=torch.zeros(G, B, C, **f), v_s=torch.zeros(G, B, C, **f),
        step=torch.zeros(G, 1, device=dev, dtype=torch.int32),
        scale_const=torch.empty(G, B, **f), lower=torch.empty(G, B, **f), upper=torch.empty(G, B, **f),
        adv=torch.zeros(G, B, 3, N, **f),  # last iterate
        bestdist=torch.empty(G, B, **f), bestscore=torch.empty(G, B, **i64),
        o_bestdist=torch.empty(G, B, **f), o_bestscore=torch.empty(G, B, **i64),
        o_bestattack=torch.zeros(G, B, 3, N, **f), pred=torch.zeros(G, B, **i64), dist_val=torch.zeros(G, B, **f),
        adv_loss=torch.zeros(G, **f), dist_loss=torch.zeros(G, **f), scaled=torch.zeros(G, **f),
        # buffers of the autograd-free iteration (_iteration_fused)
        inv_den=torch.empty(G, B, N, **f),
        gp=torch.empty(G, B, C, 3, **f), gs=torch.empty(G, B, C, **f),
        gp_reg=torch.empty(G, B, C, 3, **f), gs_reg=torch.empty(G, B, C, **f), g_adv=torch.empty(G, B, 3, N, **f),
        deform_part=torch.empty(G, ops.deform_bwd_scratch(B, N, C), **f),
        reg_scratch=torch.zeros(G, ops.regulariser_scratch(B), **f),       # zeroed: its last float is a ticket
        head_scratch=torch.zeros(G, ops.iteration_head_scratch(B, dev).numel(), **f))  # zeroed: per-cloud terms + a ticket


_STATE = ('bestdist', 'bestscore', 'o_bestdist', 'o_bestscore', 'o_bestattack', 'pred', 'dist_val')


class _Workspace:
    """Static device buffers + the captured iteration graph for one (B, N) problem shape.  ``shared`` = (stack, g): the
    buffers are group g's rows of a _Stack's (attack_many on the PointNet engine)."""

    def __init__(self, B, N, C, dev, shared=None):
        self.B, self.N, self.C = B, N, C
        bufs, g = (_buffers(1, B, N, C, dev), 0) if shared is None else (shared[0].bufs, shared[1])
        for name, t in bufs.items():
            if name not in _STATE:
                setattr(self, name, t[g])
        self.state = {name: bufs[name][g] for name in _STATE}
        if shared is None:  # the op-by-op iteration differentiates through these two
            self.P.requires_grad_()
            self.sigma.requires_grad_()
        self.graph = None       # one inner iteration
        self.graph_many = None  # `chunk` inner iterations (see HiT_ADV._chunk)
        self.chunk = 1
        self.feed = None  # pre-drawn FPS starts of a sampling victim (set per attack by _setup)

    def reset_step(self):
        for t in (self.m_p, self.v_p, self.m_s, self.v_s, self.step):
            t.zero_()
        self.state["bestdist"].fill_(1e10)
        self.state["bestscore"].fill_(-1)


class _Stack:
    """G independent attacks whose victim passes run as ONE pass over G*B clouds (HiT_ADV.attack_many on the PointNet engine).

    The victim treats clouds independently, and at B = 32 its kernels pay latency rather than throughput (the shared-layer
    chains put two 4-wave blocks on a CU, the FC stacks compute 32 rows): the pass over four stacked attacks costs 2.4x the
    pass over one (tools/victim_batch_probe.py: 301 us at B = 32, 716 us at B = 128), where four streams of B = 32 kernels
    each pay the full latency chain and serialise on the chip-filling ones.  Everything the reference couples inside a batch
    -- the normalisations of the setup phase, the batch-mean losses, the mean(scale_const) weighting, best tracking, bisection
    -- stays per attack: the three launches around the victim take a group dimension and run the per-group code on each
    group's rows (include/hitadv.h, ``*_stack``).  A group's results are the bits of an ``attack()`` call of its own
    (tests/test_gpu_attack.py::test_attack_many_equals_sequential_attacks)."""

    def __init__(self, G, B, N, C, dev):
        self.G, self.B, self.N, self.C = G, B, N, C
        self.bufs = _buffers(G, B, N, C, dev)
        self.groups = [_Workspace(B, N, C, dev, shared=(self, g)) for g in range(G)]
        self.stream = torch.cuda.Stream()
        self.graph = self.graph_many = None
        self.chunk = 1

    def all(self, name):
        """Buffer ``name`` of all groups as the kernels see it: the group dimension merged into the leading one."""
        t = self.bufs[name]
        return t.reshape(t.shape[0] * t.shape[1], *t.shape[2:]) if t.dim() > 1 else t


class _StackCaptureFailed(Exception):
    """The stacked iteration could not be warmed up / captured; ``reason`` is the original exception, ``rng_state`` the CPU
    generator's state before the stacks' setups took their draws."""

    def __init__(self, reason):
        super().__init__(repr(reason))
        self.reason, self.rng_state = reason, None


# attacks per stack in attack_many (0 / 1: no stacking, one stream per attack as before); tuning knob.  Round 4, with balanced
# stacks and round-robin launches: three stacks of 4 / 5 / 6 / 7 / 8 / 10 / 12 attacks -> 47.8 / 48.4 / 48.7 / 49.6 / 49.8-50.4 / 49.8 /
# 49.9 clouds/s (profiles/r04_sweeps.txt): eight, i.e. twenty-four attacks in flight.
_STACK = int(os.environ.get("HITADV_STACK", "8"))
# workgroups of the PointNet engine's 128 -> 1024 kernel while three or more attacks share the GPU (tuning knob; see attack_many)
_V1_BLOCKS_IN_FLIGHT = int(os.environ.get("HITADV_V1_BLOCKS_IN_FLIGHT", "128"))


class HiT_ADV:
    """Class for the HiT-ADV attack (constructor signature of the reference, :18-22)."""

    def __init__(self, model, adv_func, attack_lr=1e-2, init_weight=10., max_weight=80., binary_step=10,
                 num_iter=500, clip_func=None, cd_weight=0, curv_weight=0, ker_weight=0, hide_weight=0,
                 curv_loss_knn=32, central_num=32, total_central_num=128, max_sigm=0.7, min_sigm=0.1,
                 budget=0.1, alpha=1, use_graph='auto', verbose=True, fast_victim=True, fused_regulariser=True,
                 iterations_per_graph='auto'):
        self.model = model.cuda()
        self.model.eval()
        self.adv_func = adv_func
        self.attack_lr = attack_lr
        self.init_weight = init_weight
        self.max_weight = max_weight
        self.binary_step = binary_step
        self.num_iter = num_iter
        self.clip_func = clip_func  # stored, never used (as in the reference)
        self.cd_weight = cd_weight
        self.curv_weight = curv_weight  # stored, never used
        self.hide_weight = hide_weight
        self.ker_weight = ker_weight
        self.curv_loss_knn = curv_loss_knn
        self.central_num = central_num
        self.max_sigm = max_sigm
        self.min_sigm = min_sigm
        self.budget = budget
        self.alpha = alpha
        self.total_central_num = total_central_num
        self.use_graph = use_graph
        self.verbose = verbose
        self.fast_victim = fast_victim
        self.fused_regulariser = fused_regulariser
        self.iterations_per_graph = iterations_per_graph  # 'auto': what the victim's view asks for (PointNet engine: 10)
        self._view = None
        self.attacks_per_stack = _STACK  # attack_many: victim passes of this many attacks merged into one (PointNet engine)
        self._chamfer = ChamferDist()
        self._ws = {}
        self.last_graph_used = False

    # ------------------------------------------------------------------ small pieces
    def _victim(self):
        """The callable used for forward/backward.  Victims that offer ``attack_view()`` (our PointNet:
        eval-mode BatchNorm folded, points-major GEMMs) are run through that view; its buffers are
        re-folded in place at every attack() so weight updates are seen and the captured graph stays valid."""
        if not (self.fast_victim and hasattr(self.model, 'attack_view')):
            return self.model
        if self._view is None:
            try:
                self._view = self.model.attack_view()
            except NotImplementedError:  # a configuration the view does not cover: the module itself is the victim
                self.fast_victim = False
                return self.model
        return self._view

    def _logits(self, x, feed=None):
        """Victim forward; ``feed`` = this attack's pre-drawn FPS starts for a victim that samples (model/_sampling.py)."""
        with _sampling.using(feed):
            out = self._victim()(x)
        return out[0] if isinstance(out, tuple) else out

    def get_gradient(self, data, target):
        """d CE / d xyz for data[B,3,K]; returns (grad, number of clean misclassifications). (:537-559)"""
        x = data.clone().detach().float().cuda().requires_grad_()
        target = target.long().cuda()
        logits = self._logits(x)
        grad, = torch.autograd.grad(F.cross_entropy(logits, target), x)
        miss = (logits.argmax(dim=-1) != target).sum().item()
        return grad.detach(), miss

    def kernel_density(self, central_points, pc, delta):
        """[B,C,N] kernel matrix exp(-|x-c| / (2 delta^2)) (:298-304); diagnostic helper only --
        the attack itself never materialises it."""
        diff = pc.unsqueeze(2) - central_points.unsqueeze(3)  # [B,3,C,N]
        return torch.exp(-diff.norm(dim=1) / (2 * delta * delta).unsqueeze(2))

    def transformation_loss(self, adv_data, perturb_mat, gauss_delta, batch_avg=True):
        """(|P| + |1 - sigma|) / C over the whole batch tensor or per sample (:306-316)."""
        if batch_avg:
            t = torch.norm(perturb_mat) + torch.norm(1 - gauss_delta)
        else:
            t = torch.norm(perturb_mat, dim=(1, 2)) + torch.norm(1 - gauss_delta, dim=1)
        return t / self.central_num

    def curv_std_loss(self, gauss_delta, central_kappa_std, max_delta, min_delta):
        """Cosine similarity between normalised centre curvature-std and normalised sigma (:341-346)."""
        lo, hi = central_kappa_std.min(), central_kappa_std.max()
        ref = ((central_kappa_std - lo) / (hi - lo + 1e-7)).squeeze(-1)
        return self._hide(gauss_delta, ref, max_delta, min_delta)

    @staticmethod
    def _hide(gauss_delta, ref, max_delta, min_delta):
        return F.cosine_similarity(ref, (gauss_delta - min_delta) / (max_delta - min_delta + 1e-7))

    def farthest_point_sample(self, xyz, npoint):
        """FPS with a random start drawn from the CPU generator, xyz[B,N,3] -> [B,npoint] (:489-510)."""
        B, N, _ = xyz.shape
        start = torch.randint(0, N, (B,), dtype=torch.long)
        return ops.fps_from_start(xyz, npoint, start.to(xyz.device))

    # ------------------------------------------------------------------ setup phase
    @torch.no_grad()
    def _select_centres(self, ori, normal, grad):
        """Saliency + curvature scoring, FPS seeds, best-scoring neighbour per seed, top-C (:61-93,118-123)."""
        k = self.curv_loss_knn
        B = ori.shape[0]
        kstd, kappa, _ = curvature_std(ori, normal, k)
        centre = torch.median(ori, dim=-1)[0]
        off = ori - centre[:, :, None]
        r = torch.sum(off ** 2, dim=1) ** 0.5
        sal = -1. * (r ** self.alpha) * torch.sum(off * grad, dim=1)
        sal_n = (sal - sal.min()) / (sal.max() - sal.min() + 1e-7)
        std_n = (kstd - kstd.min()) / (kstd.max() - kstd.min() + 1e-7)
        score = 0.001 * sal_n + std_n  # [B,N]

        pts = ori.transpose(1, 2).contiguous()
        far_idx = self.farthest_point_sample(pts, self.total_central_num)
        nbr_idx = knn_points(_take(pts, far_idx), pts, K=k + 1).idx  # [B,T,k+1]
        T = nbr_idx.shape[1]
        nbr_score = knn_gather(score.unsqueeze(2), nbr_idx).squeeze(-1)  # [B,T,k+1]
        pick = nbr_score.topk(k=1, dim=2)[1]  # [B,T,1]
        cand_idx = nbr_idx.gather(2, pick).squeeze(-1)  # [B,T] point index of every candidate
        cand_score = nbr_score.gather(2, pick).squeeze(-1)
        _, top = torch.topk(cand_score, k=self.central_num)  # [B,C]
        central_idx = cand_idx.gather(1, top)
        central = _take(pts, central_idx).transpose(1, 2).contiguous()  # [B,3,C]
        central_kappa = kappa.gather(1, central_idx).unsqueeze(-1)  # [B,C,1]
        return central, central_kappa, score

    # ------------------------------------------------------------------ one inner iteration
    def _iteration(self, ws):
        """Everything between two Adam steps (:156-246), host-sync free."""
        if self.fused_regulariser and hasattr(self.adv_func, 'fused'):
            return self._iteration_fused(ws)
        with torch.no_grad():
            ws.P.clamp_(-self.budget, self.budget)
            ws.sigma.clamp_(self.min_sigm, self.max_sigm)
        adv = ops.deform(ws.ori, ws.central, ws.P, ws.sigma)
        logits = self._logits(adv, ws.feed)
        ops.best_update(logits.detach(), ws.target, ws.P.detach(), ws.sigma.detach(), adv.detach(), ws.state)

        adv_loss = self.adv_func(logits, ws.target)
        regs = (self.cd_weight, self.ker_weight, self.hide_weight)
        if self.fused_regulariser and any(w != 0 for w in regs):
            # one autograd node: the three regularisers, their gradients and the mean(scale_const) weighting
            scaled = ops.regulariser(ws.P, ws.sigma, adv, ws.ori, ws.hide_ref, ws.scale_const, regs,
                                     (self.min_sigm, self.max_sigm), ws.dist_loss)
            loss = adv_loss + scaled
            dist_loss = None
        else:
            dist_loss = torch.zeros((), device=adv.device)
            if self.cd_weight != 0:
                # quirk Q1 kept: the operator receives [B,3,N] tensors (:230)
                w = torch.full((ws.B,), float(self.cd_weight), device=adv.device)
                dist_loss = dist_loss + self._chamfer(adv, ws.ori, w)
            if self.ker_weight != 0:
                dist_loss = dist_loss + self.transformation_loss(adv, ws.P, ws.sigma) * self.ker_weight
            if self.hide_weight != 0:
                hide = self._hide(ws.sigma, ws.hide_ref, self.max_sigm, self.min_sigm) * self.hide_weight
                dist_loss = dist_loss + hide.mean()
            loss = (adv_loss + ws.scale_const * dist_loss).mean()
        g_p, g_s = torch.autograd.grad(loss, [ws.P, ws.sigma])
        ops.adam_step(ws.P, ws.sigma, g_p, g_s, ws.m_p, ws.v_p, ws.m_s, ws.v_s, ws.step,
                      self.attack_lr * 5, self.attack_lr * 3)
        ws.adv.copy_(adv.detach())
        ws.adv_loss.copy_(adv_loss.detach())
        if dist_loss is not None:
            ws.dist_loss.copy_(dist_loss.detach())

    def _iteration_fused(self, ws):
        """The same iteration with the chain rule written out instead of recorded: autograd is used for the victim
        only (any nn.Module), every other forward / backward is an explicit kernel call on workspace buffers, and
        gradient sums ride inside the consuming kernels (victim + regulariser into deform_bwd's upstream; deformation +
        regulariser into Adam).  4 launches around the victim instead of ~37 (deform_fwd, iteration_head with the
        regularisers' forward pass, deform_bwd and Adam with the regularisers' backward terms).  The projection of (perturb, sigma)
        (:157-158) is applied by the Adam kernel right after the update -- the parameters every forward pass sees
        are the same (the initial draws already lie inside the box)."""
        regs = (self.cd_weight, self.ker_weight, self.hide_weight)
        rng = (self.min_sigm, self.max_sigm)
        P, sigma = ws.P.detach(), ws.sigma.detach()
        view = self._view
        # the PointNet engine can deform the cloud in its own first kernel (same bits as deform_fwd): one launch less
        own_deform = (getattr(view, 'hip_engine', False) and hasattr(view, 'deform_inputs') and ws.C <= 256
                      and getattr(view, 'fold_small_layers', False))
        if not own_deform:
            ops.deform_fwd_into(ws.ori, ws.central, P, sigma, ws.adv, ws.inv_den)
        x = ws.adv.detach().requires_grad_()
        fused = hasattr(self.adv_func, 'fused_kind')
        # ... and leave its last layer (256 -> classes) to the loss kernel below: another one
        defer = (fused and any(w != 0 for w in regs) and getattr(view, 'hip_engine', False) and hasattr(view, 'defer_logits')
                 and view.h3_w.shape[0] <= 256 and view.h3_w.shape[1] <= 64)
        head = None
        if own_deform:
            view.deform_inputs = (ws.ori, ws.central, P, sigma, ws.inv_den)
        if defer:
            view.defer_logits = True
        try:
            logits = self._logits(x, ws.feed)
            head = view.pending_head if defer else None
            if own_deform and view.deform_inputs is not None:  # the forward pass did not go through the engine's kernel
                raise RuntimeError("the victim was asked to deform the cloud in its first kernel and did not")
        finally:
            if own_deform:
                view.deform_inputs = None
            if defer:
                view.defer_logits, view.pending_head = False, None
        reg_done = False
        if hasattr(self.adv_func, 'fused_kind'):  # best-result tracking + adversarial loss: one launch
            kind, kappa = self.adv_func.fused_kind()
            dlogits = torch.empty_like(logits)
            if any(w != 0 for w in regs):  # ... which also takes the regularisers' forward pass (they need no victim output)
                ops.iteration_head_reg(logits.detach(), ws.target, P, sigma, ws.adv, ws.state, ws.step, kind, kappa,
                                       ws.adv_loss, dlogits, ws.head_scratch, ws.ori, ws.hide_ref, ws.scale_const, regs, rng,
                                       ws.reg_scratch, ws.dist_loss, ws.scaled, head=head)
                reg_done = True
            else:
                ops.iteration_head(logits.detach(), ws.target, P, sigma, ws.adv, ws.state, ws.step, kind, kappa,
                                   ws.adv_loss, dlogits, ws.head_scratch)
        else:
            ops.best_update(logits.detach(), ws.target, P, sigma, ws.adv, ws.state, counter=ws.step)
            _, dlogits = self.adv_func.fused(logits, ws.target, loss_out=ws.adv_loss)
        g_victim, = torch.autograd.grad(logits, x, grad_outputs=dlogits)
        clamp_p = (-self.budget, self.budget)
        if any(w != 0 for w in regs):
            if not reg_done:
                ops.regulariser_fwd_fused_into(P, sigma, ws.adv, ws.ori, ws.hide_ref, ws.scale_const, regs, rng,
                                               ws.reg_scratch, ws.dist_loss, ws.scaled)
            # the regularisers' backward terms are closed-form in what the next two kernels read anyway: evaluated inside
            # them (same bits as regulariser_bwd_add), no launch of their own.  (ops.deform_bwd_adam_reg runs the Adam step
            # as the tail of the deformation's backward kernel, one launch for both: measured 4 us SLOWER per iteration --
            # the hand-off and the last block's serial tail cost more than the kernel boundary -- so two launches stay.)
            ops.deform_bwd_partials_reg_into(ws.ori, ws.central, P, sigma, ws.adv, ws.inv_den, g_victim.contiguous(),
                                             ws.reg_scratch, regs, ws.deform_part)
            ops.adam_step_partials_reg(P, sigma, ws.deform_part, ws.N, ws.hide_ref, ws.reg_scratch, regs, rng, ws.m_p, ws.v_p,
                                       ws.m_s, ws.v_s, ws.step, self.attack_lr * 5, self.attack_lr * 3, clamp_p, rng)
        else:
            # the deformation's gradient stays in its per-slab partials; the Adam kernel sums them (in the reduce order)
            ops.deform_bwd_partials_into(ws.ori, ws.central, P, sigma, ws.adv, ws.inv_den, g_victim.contiguous(),
                                         ws.deform_part)
            ops.adam_step_partials(P, sigma, ws.deform_part, ws.N, None, None, ws.m_p, ws.v_p, ws.m_s, ws.v_s, ws.step,
                                   self.attack_lr * 5, self.attack_lr * 3, clamp_p, rng)

    def _warm_up(self, ws):
        """Two eager passes of the iteration on ``ws.stream``; the second under PyTorch's sync-debug mode set
        to "error": anything that would synchronise with the host (an adv_func calling .item(), a victim
        drawing a CPU randint and copying it over, ...) raises HERE, in eager mode.  Attempting a capture with
        such an op inside would invalidate it and can leave PyTorch's capture bookkeeping (RNG registration,
        current stream) in a broken state.  Returns None when capturable, else the reason."""
        reason = None
        prev_mode = torch.cuda.get_sync_debug_mode()
        try:
            ws.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ws.stream):
                if ws.feed is not None:
                    ws.feed.seek(0)
                self._iteration(ws)  # unguarded: lets library handles / lazy initialisation happen
                if ws.feed is not None:
                    ws.feed.seek(0)  # both passes read row 0: a one-iteration attack has no row 1
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")  # "prototype feature" notice
                    torch.cuda.set_sync_debug_mode("error")
                self._iteration(ws)
        except Exception as e:  # noqa: BLE001
            reason = e
        finally:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.cuda.set_sync_debug_mode(prev_mode)
            torch.cuda.current_stream().wait_stream(ws.stream)
            torch.cuda.synchronize()
        return reason

    def _chunk(self):
        """Inner iterations recorded into the second, longer hipGraph of a workspace.  A graph launch costs the host
        ~0.1 ms for the ~40 kernels of a PointNet iteration; with three attacks in flight that is the whole iteration
        time, i.e. the host -- not the GPU -- paces the loop.  Replaying u iterations per launch divides that cost by u;
        the single-iteration graph serves the iterations whose state the host reads back (progress lines) and remainders."""
        want = self.iterations_per_graph
        if want == 'auto':
            want = getattr(self._view, 'iterations_per_graph', 1) if self._view is not None else 1
        return max(1, min(int(want), self.num_iter))

    def _prepare_graphs(self, wss):
        """Warm up every workspace, THEN capture one hipGraph per workspace (``_iteration`` on its stream).

        Order matters on this stack (ROCm 7.0 runtime bundled with PyTorch 2.10): a replay that follows eager
        victim work issued after the capture faults with HSA_STATUS_ERROR_EXCEPTION 0x1016 for some graphs
        (measured).  So all eager work of all workspaces comes first, graphs are captured
        last, and they are not kept across attack() calls (capturing costs ~0.1 s against seconds of replays)."""
        for ws in wss:
            ws.graph = ws.graph_many = None
        if self.use_graph in (False, 'never') or self.num_iter * self.binary_step == 0:
            return
        reason = None
        for ws in wss:
            reason = reason or self._warm_up(ws)
        if reason is None:
            try:
                chunk = self._chunk()
                for ws in wss:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=ws.stream):
                        self._iteration(ws)
                    ws.graph, ws.graph_many, ws.chunk = g, None, 1
                    if chunk > 1:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=ws.stream):
                            for _ in range(chunk):
                                self._iteration(ws)
                        ws.graph_many, ws.chunk = g, chunk
                return
            except Exception as e:  # noqa: BLE001
                reason = e
                torch.cuda.synchronize()
                for ws in wss:
                    ws.graph = ws.graph_many = None
        if self.use_graph is True or self.use_graph == 'always':
            raise RuntimeError("the HiT-ADV iteration cannot be captured into a hipGraph: %r" % (reason,))
        warnings.warn("the HiT-ADV iteration is not hipGraph-capturable (%r); running the eager loop" % (reason,))

    # ------------------------------------------------------------------ attack phases
    def _setup(self, data, target, slot=0, into=None):
        """Everything before the binary search (:51-123): scoring, centre selection, state initialisation, and
        ALL random draws of this attack, taken from the global CPU generator in the reference's order (randint for
        the FPS start :501, then per binary step rand(B,C,3) :130 and rand(B,C) :133) and uploaded once."""
        B, K = data.shape[:2]
        dev = torch.device('cuda', torch.cuda.current_device())
        ori = data[:, :, :3].float().to(dev).clone().detach().transpose(1, 2).contiguous()
        normal = data[:, :, 3:].float().to(dev).clone().detach().transpose(1, 2).contiguous()
        target = target.long().to(dev).detach()
        C = self.central_num
        grad, _ = self.get_gradient(ori, target)
        central, central_kappa, _ = self._select_centres(ori, normal, grad)

        if into is not None:
            ws = into
            if (ws.B, ws.N, ws.C) != (B, K, C):
                raise RuntimeError("stacked attacks need batches of one shape: got (%d, %d), the stack holds (%d, %d)"
                                   % (B, K, ws.B, ws.N))
        else:
            key = (B, K, C, slot)
            ws = self._ws.get(key)
            if ws is None:
                ws = self._ws[key] = _Workspace(B, K, C, dev)
                ws.stream = torch.cuda.Stream()
        ws.ori.copy_(ori)
        ws.central.copy_(central)
        ws.target.copy_(target)
        lo, hi = central_kappa.min(), central_kappa.max()
        ws.hide_ref.copy_(((central_kappa - lo) / (hi - lo + 1e-7)).squeeze(-1))
        ws.scale_const.fill_(self.init_weight)
        # A victim that samples (PointNet++, PCT) draws its FPS starts from the same CPU generator in every forward
        # pass: per binary step the reference's order is rand(B,C,3), rand(B,C), then num_iter forward passes' worth of
        # randint draws.  Taking them here, in that order, keeps a seeded run on the reference's trajectory and leaves
        # nothing in the iteration that needs the host (the iteration becomes capturable).
        highs = _sampling.plan_of(self.model, K)
        draws_p, draws_s, starts = [], [], []
        for _ in range(self.binary_step):
            draws_p.append(torch.rand(B, C, 3) * torch.tensor(self.budget))
            draws_s.append(torch.rand((B, C)))
            if highs:
                starts.append(_sampling.StartFeed.draw(highs, B, self.num_iter))
        ws.rand_P = torch.stack(draws_p).to(dev) if draws_p else None
        ws.rand_S = torch.stack(draws_s).to(dev) if draws_s else None
        ws.feed = (_sampling.StartFeed(highs, B, self.binary_step * self.num_iter, dev, table=torch.cat(starts))
                   if starts else None)
        return ws

    def _reset_search(self, ws):
        ws.lower.zero_()
        ws.upper.fill_(self.max_weight)
        ws.scale_const.fill_(self.init_weight)
        st = ws.state
        st["o_bestdist"].fill_(1e10)
        st["o_bestscore"].fill_(-1)
        st["o_bestattack"].zero_()
        ws.adv_loss.zero_()
        ws.dist_loss.zero_()

    def _run_step(self, ws, binary_step, verbose):
        """One binary-search step (:125-273) enqueued on the current stream: fresh parameters, num_iter replays
        (or eager iterations), per-sample bisection of the distance weight -- no host synchronisation unless
        ``verbose`` asks for the reference's progress lines."""
        B, st = ws.B, ws.state
        self._begin_step(ws, binary_step)
        report_every = max(1, self.num_iter // 5)
        iteration = 0
        while iteration < self.num_iter:
            report = verbose and iteration % report_every == 0
            if report:
                prev = (ws.adv_loss.item(), ws.dist_loss.item())
            # iterations that may run before the host next looks at the state: a reported iteration runs on its own
            until = self.num_iter if not verbose else min(self.num_iter, (iteration // report_every + 1) * report_every)
            many = (not report) and ws.graph_many is not None and until - iteration >= ws.chunk
            if many:
                ws.graph_many.replay()
            elif ws.graph is not None:
                ws.graph.replay()
            else:
                self._iteration(ws)
            if report:
                success_num = (st["pred"] != ws.target).sum().item()
                print('Step {}, iteration {}, success {}/{}\n'
                      'adv_loss: {:.4f}, dist_loss: {:.4f}'.format(binary_step, iteration, success_num, B,
                                                                   prev[0], prev[1]))
            iteration += ws.chunk if many else 1
        self._end_step(ws)

    def _begin_step(self, ws, binary_step):
        """Fresh parameters of a binary-search step from the draws taken at setup (:128-141), Adam and per-step best state reset."""
        B, C = ws.B, ws.C
        if ws.feed is not None:
            ws.feed.seek(binary_step * self.num_iter)  # this step's rows (the warm-up passes moved the cursor)
        with torch.no_grad():
            ws.P.copy_(ws.rand_P[binary_step])
            ws.sigma.copy_(torch.ones((B, C), device=ws.P.device) * self.min_sigm
                           + ws.rand_S[binary_step] * (self.max_sigm - self.min_sigm))
        ws.reset_step()
        ws.adv_loss.zero_()
        ws.dist_loss.zero_()

    def _end_step(self, ws):
        """Per-sample bisection of the distance weight (:264-273), on the device."""
        st = ws.state
        with torch.no_grad():
            ok = ((st["bestscore"] != ws.target) & (st["bestscore"] != -1)
                  & (st["bestdist"] <= st["o_bestdist"]))
            ws.lower.copy_(torch.where(ok, torch.maximum(ws.lower, ws.scale_const), ws.lower))
            ws.upper.copy_(torch.where(ok, ws.upper, torch.minimum(ws.upper, ws.scale_const)))
            ws.scale_const.copy_((ws.lower + ws.upper) / 2.)

    # ------------------------------------------------------------------ stacked attacks (attack_many on the PointNet engine)
    def _can_stack(self):
        """Stacking needs the iteration whose victim pass is ONE call that takes the deformation and hands back the classifier
        head's input: the PointNet engine under the fused adversarial loss with at least one regulariser on."""
        view = self._victim()
        return (view is self._view and getattr(view, 'hip_engine', False) and hasattr(view, 'deform_inputs')
                and getattr(view, 'fold_small_layers', False) and hasattr(view, 'defer_logits')
                and self.fused_regulariser and hasattr(self.adv_func, 'fused_kind') and self.central_num <= 256
                and any(w != 0 for w in (self.cd_weight, self.ker_weight, self.hide_weight))
                and view.h3_w.shape[0] <= 256 and view.h3_w.shape[1] <= 64 and self.use_graph not in (False, 'never'))

    def stacks(self):
        """Whether ``attack_many`` will really merge the victim passes of its attacks (the knob is on AND the victim / loss
        allow it): what callers that size groups of attacks have to ask -- not the knob alone."""
        return self.attacks_per_stack > 1 and self._can_stack()

    # one stream per attack: four measured best, odd counts worst; HITADV_UNSTACKED_IN_FLIGHT: tuning knob
    UNSTACKED_IN_FLIGHT = int(os.environ.get("HITADV_UNSTACKED_IN_FLIGHT", "4"))

    def in_flight(self, requested):
        """Attacks to hand to ``attack_many`` at a time: ``requested`` where the victim passes are stacked (PointNet engine:
        24 by default = three balanced stacks of up to eight, ``hit_adv_amd.stack_sizes``), at most ``UNSTACKED_IN_FLIGHT`` otherwise -- every un-stacked attack in flight holds
        a workspace, a stream and the victim's activations of its own (DGCNN, PointNet++, PCT)."""
        requested = max(1, int(requested))
        return requested if self.stacks() else min(requested, self.UNSTACKED_IN_FLIGHT)

    def _iteration_stacked(self, stack):
        """``_iteration_fused`` for G attacks at once: one victim forward pass over the G*B clouds (its first kernel deforms
        them, its last layer is left to the loss kernel), the loss / best-tracking / regulariser launch for all groups, one
        victim backward pass, the deformation's backward and the Adam step for all groups -- 32 launches for G attacks."""
        regs = (self.cd_weight, self.ker_weight, self.hide_weight)
        rng = (self.min_sigm, self.max_sigm)
        view, G, A = self._view, stack.G, stack.all
        kind, kappa = self.adv_func.fused_kind()
        x = A('adv').detach().requires_grad_()
        view.deform_inputs = (A('ori'), A('central'), A('P'), A('sigma'), A('inv_den'))
        view.defer_logits = True
        try:
            logits = self._logits(x)
            head = view.pending_head
            if view.deform_inputs is not None:
                raise RuntimeError("the victim was asked to deform the cloud in its first kernel and did not")
        finally:
            view.deform_inputs = None
            view.defer_logits, view.pending_head = False, None
        dlogits = torch.empty_like(logits)
        state = {name: A(name) for name in _STATE}
        ops.iteration_head_reg(logits.detach(), A('target'), A('P'), A('sigma'), A('adv'), state, A('step'), kind, kappa,
                               A('adv_loss'), dlogits, A('head_scratch'), A('ori'), A('hide_ref'), A('scale_const'), regs, rng,
                               A('reg_scratch'), A('dist_loss'), A('scaled'), head=head, groups=G)
        g_victim, = torch.autograd.grad(logits, x, grad_outputs=dlogits)
        clamp_p = (-self.budget, self.budget)
        ops.deform_bwd_partials_reg_into(A('ori'), A('central'), A('P'), A('sigma'), A('adv'), A('inv_den'),
                                         g_victim.contiguous(), A('reg_scratch'), regs, A('deform_part'), groups=G)
        ops.adam_step_partials_reg(A('P'), A('sigma'), A('deform_part'), stack.N, A('hide_ref'), A('reg_scratch'), regs, rng,
                                   A('m_p'), A('v_p'), A('m_s'), A('v_s'), A('step'), self.attack_lr * 5, self.attack_lr * 3,
                                   clamp_p, rng, groups=G)

    def _prepare_stack_graphs(self, stacks):
        """Two warm-up passes per stack (the second under sync-debug "error"), then one- and many-iteration graphs.  A failure
        of either leaves no graph and no half-set view state behind and is reported as ``_StackCaptureFailed``."""
        try:
            self._prepare_stack_graphs_unguarded(stacks)
        except Exception as e:  # noqa: BLE001
            torch.cuda.synchronize()
            for st in stacks:
                st.graph = st.graph_many = None
            view = self._view
            if view is not None:  # what _iteration_stacked sets for the duration of one victim call
                view.deform_inputs = None
                view.defer_logits, view.pending_head = False, None
            raise _StackCaptureFailed(e)

    def _prepare_stack_graphs_unguarded(self, stacks):
        prev_mode = torch.cuda.get_sync_debug_mode()
        for st in stacks:
            st.graph = st.graph_many = None
            try:
                st.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st.stream):
                    self._iteration_stacked(st)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        torch.cuda.set_sync_debug_mode("error")
                    self._iteration_stacked(st)
            finally:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    torch.cuda.set_sync_debug_mode(prev_mode)
                torch.cuda.current_stream().wait_stream(st.stream)
                torch.cuda.synchronize()
        chunk = self._chunk()
        for st in stacks:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st.stream):
                self._iteration_stacked(st)
            st.graph, st.graph_many, st.chunk = g, None, 1
            if chunk > 1:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st.stream):
                    for _ in range(chunk):
                        self._iteration_stacked(st)
                st.graph_many, st.chunk = g, chunk

    def _replay_round_robin(self, units):
        """``num_iter`` iterations of every unit (stacks, or workspaces of un-stacked attacks: anything with ``stream``,
        ``graph``, ``graph_many``, ``chunk``), the host going ROUND the units one graph launch at a time.  The runtime lets the
        host run only so far ahead of the GPU; queueing one unit's whole binary step first would leave the other streams
        empty for most of it (measured on cfg5's CW sweep: 19.7 s attack after attack, 14.1 s going round)."""
        it = 0
        while it < self.num_iter:
            many = all(u.graph_many is not None for u in units) and self.num_iter - it >= max(u.chunk for u in units)
            for u in units:
                with torch.cuda.stream(u.stream):
                    if many:
                        u.graph_many.replay()
                    elif u.graph is not None:
                        u.graph.replay()
                    else:
                        self._iteration(u)
            it += units[0].chunk if many else 1

    def _attack_stacked(self, batches, per_stack):
        """attack_many through stacks of ``per_stack`` attacks: every stack on its own stream, its victim passes merged."""
        B, K = batches[0][0].shape[:2]
        dev = torch.device('cuda', torch.cuda.current_device())
        from .. import stack_sizes
        sizes = stack_sizes(len(batches), per_stack)
        rng_before = torch.get_rng_state()  # the setups below draw from the global CPU generator
        stacks, i = [], 0
        for n, G in enumerate(sizes):
            key = (B, K, self.central_num, 'stack', n, G)
            st = self._ws.get(key)
            if st is None:
                st = self._ws[key] = _Stack(G, B, K, self.central_num, dev)
            for ws, (d, t) in zip(st.groups, batches[i:i + G]):  # setup in batch order: the reference's order of random draws
                self._setup(d, t, into=ws)
            stacks.append(st)
            i += G
        try:
            self._prepare_stack_graphs(stacks)
        except _StackCaptureFailed as e:
            e.rng_state = rng_before
            raise
        self.last_graph_used = True
        for st in stacks:
            for ws in st.groups:
                self._reset_search(ws)
            st.stream.wait_stream(torch.cuda.current_stream())
        for binary_step in range(self.binary_step):
            for st in stacks:
                with torch.cuda.stream(st.stream):
                    for ws in st.groups:
                        self._begin_step(ws, binary_step)
            self._replay_round_robin(stacks)
            for st in stacks:
                with torch.cuda.stream(st.stream):
                    for ws in st.groups:
                        self._end_step(ws)
        for st in stacks:
            torch.cuda.current_stream().wait_stream(st.stream)
        return [self._finish(ws, False) for st in stacks for ws in st.groups]

    def _finish(self, ws, verbose):
        """Failure fill and return value (:277-287)."""
        st = ws.state
        with torch.no_grad():
            fail = ws.lower == 0.
            best = torch.where(fail[:, None, None], ws.adv, st["o_bestattack"])
            st["o_bestdist"].copy_(torch.where(fail, st["dist_val"], st["o_bestdist"]))
        lower_cpu = ws.lower.cpu()
        if hasattr(self._view, 'check_range'):
            self._view.check_range()  # fp16x2 victim layers: loud if an operand left fp16's range
        from ..model import _pointwise
        _pointwise.check_range(ws.adv.device)
        self.last_lower_bound = lower_cpu
        self.last_bestdist = st["o_bestdist"].cpu()
        success_num = (lower_cpu > 0.).sum()
        if verbose:
            print('lower_bound is', lower_cpu)
            print('Successfully attack {}/{}'.format(success_num, ws.B))
        return best.double().cpu().numpy().transpose((0, 2, 1)), success_num

    @degrade_on_fp16_range
    def attack(self, data, target):
        """Attack on given data to target.

        Args:
            data (torch.FloatTensor): victim data with normals, [B, num_points, 6]
            target (torch.LongTensor): true labels (the attack is untargeted), [B]
        Returns:
            (numpy.float64 [B, num_points, 3], 0-d torch.int64 tensor with the success count)
        """
        if self._view is not None:
            self._view.refresh(self.model)
        ws = self._setup(data, target)
        self._prepare_graphs([ws])
        self.last_graph_used = ws.graph is not None
        self._reset_search(ws)
        ws.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws.stream):
            for binary_step in range(self.binary_step):
                self._run_step(ws, binary_step, self.verbose)
        torch.cuda.current_stream().wait_stream(ws.stream)
        return self._finish(ws, self.verbose)

    @degrade_on_fp16_range
    def attack_many(self, batches):
        """Attack several independent batches CONCURRENTLY on one GPU: ``[(data, target), ...] -> [(adv, n), ...]``.

        Results are those of calling ``attack`` on the batches one after the other (same RNG draws in the same
        order, same kernels per batch) -- every batch keeps the reference's per-call semantics -- but the
        binary-search steps of the batches are enqueued on separate HIP streams, each replaying its own captured
        graph.  One B=32 PointNet iteration is ~115 short kernels that leave most of the 256 CUs idle between
        launches; a second, independent stream fills those gaps.  No progress lines are printed."""
        if self._view is not None:
            self._view.refresh(self.model)
        # with three or more attacks in flight the victim's 128 -> 1024 layers run on half the chip each (twice as long):
        # the other half stays free for the other streams' short kernels (bench.py: 28.1 instead of 27.0 clouds/s at four)
        self._victim()  # the view is created on first use: it has to exist before its grid is chosen
        per_stack = self.attacks_per_stack
        if (per_stack > 1 and len(batches) > 1 and self._can_stack()
                and len({tuple(d.shape[:2]) for d, _ in batches}) == 1):
            view = self._view
            before, view.linear_max_blocks = view.linear_max_blocks, (
                _V1_BLOCKS_IN_FLIGHT if len(batches) > per_stack else view.linear_max_blocks)  # two stacks or more in flight
            try:
                return self._attack_stacked(batches, per_stack)
            except _StackCaptureFailed as e:
                # as _prepare_graphs does for the un-stacked path: 'auto' warns and takes the path that needs no stacked graph
                # (one stream per attack, each with its own capture attempt and eager fallback); True / 'always' is loud
                if self.use_graph is True or self.use_graph == 'always':
                    raise RuntimeError("the stacked HiT-ADV iteration cannot be captured into a hipGraph: %r" % (e.reason,))
                warnings.warn("the stacked HiT-ADV iteration is not hipGraph-capturable (%r); attacking the batches on "
                              "separate streams instead" % (e.reason,))
                torch.set_rng_state(e.rng_state)  # the batches' draws are taken again, in the same order
            finally:
                view.linear_max_blocks = before
        view = self._view if hasattr(self._view, 'linear_max_blocks') else None
        if view is not None:
            before, view.linear_max_blocks = view.linear_max_blocks, (_V1_BLOCKS_IN_FLIGHT if len(batches) >= 3 else view.linear_max_blocks)
        try:  # 64 workgroups (two clouds per block): 27.3, 32: 22.0 clouds/s
            wss = [self._setup(d, t, slot=i) for i, (d, t) in enumerate(batches)]
            self._prepare_graphs(wss)
            self.last_graph_used = all(ws.graph is not None for ws in wss)
            for ws in wss:
                self._reset_search(ws)
                ws.stream.wait_stream(torch.cuda.current_stream())
            same_chunk = len({ws.chunk for ws in wss}) == 1
            for binary_step in range(self.binary_step):
                if not same_chunk:  # (never in practice: the chunk is a property of the attacker, not of a workspace)
                    for ws in wss:
                        with torch.cuda.stream(ws.stream):
                            self._run_step(ws, binary_step, False)
                    continue
                for ws in wss:
                    with torch.cuda.stream(ws.stream):
                        self._begin_step(ws, binary_step)
                self._replay_round_robin(wss)
                for ws in wss:
                    with torch.cuda.stream(ws.stream):
                        self._end_step(ws)
            for ws in wss:
                torch.cuda.current_stream().wait_stream(ws.stream)
            return [self._finish(ws, False) for ws in wss]
        finally:
            if view is not None:
                view.linear_max_blocks = before
