"""Targeted CW point-perturbation attack, interface of the reference's CW/PerturbT.py::CWPerturbT (ctor :16-42,
attack :44-183): CW/Perturb.py without the ``pre_head`` hook and without the channel-layout sniffing -- the input is
always [B, num_points, 3]."""
from .Perturb import CWPerturb


class CWPerturbT(CWPerturb):
    """Class for CW attack (targeted)."""

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, init_weight=10., max_weight=80., binary_step=10,
                 num_iter=500, clip_func=None, verbose=True, fast_victim=True, use_graph='auto'):
        super().__init__(model, adv_func, dist_func, attack_lr=attack_lr, init_weight=init_weight,
                         max_weight=max_weight, binary_step=binary_step, num_iter=num_iter, pre_head=None,
                         clip_func=clip_func, verbose=verbose, fast_victim=fast_victim, use_graph=use_graph)

    def attack(self, data, target):
        """data [B,num_points,3], target [B] -> (float64 ndarray [B,num_points,3], successes)."""
        return super().attack(data.transpose(1, 2).contiguous(), target, _channel_first=True)
