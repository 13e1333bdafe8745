"""Untargeted AOF + auto-encoder attack, interface of the reference's CW/UAEAOF.py::CWUAEAOF (ctor :58-83, attack :85-241)."""
from ._family import _CWFamily
from .AOF import get_Laplace_from_pc, knn  # noqa: F401


class CWUAEAOF(_CWFamily):
    """Class for the AOF attack with an additional auto-encoder view; loss weights (1-2*GAMMA, GAMMA, GAMMA)."""
    spectral = True
    fresh = False  # :180-183

    def __init__(self, model, ae_model, adv_func, dist_func, attack_lr=1e-2, binary_step=2, num_iter=200, GAMMA=0.25,
                 low_pass=100, clip_func=None, verbose=True, fast_victim=True, use_graph='auto'):
        self.fast_victim, self.use_graph = fast_victim, use_graph
        self._setup(model, adv_func, dist_func, attack_lr, binary_step, num_iter, GAMMA, clip_func, verbose,
                    ae_model=ae_model, low_pass=low_pass)

    def attack(self, data, target):
        return self._run(data, target)
