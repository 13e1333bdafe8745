"""Targeted AOF attack, interface of the reference's CW/TAOF.py::CWTAOF (ctor :58-81, attack :83-242)."""
from ._family import _CWFamily
from .AOF import get_Laplace_from_pc, knn  # noqa: F401  (module-level names of the reference file)


class CWTAOF(_CWFamily):
    """Class for the targeted AOF attack: ``pred == target`` on the full cloud, low-frequency view away from ``y_truth``."""
    spectral = True
    targeted = True
    fresh = True
    final_clip = False   # the reference comments the final clip out (:232)
    freeze_model = True  # :105-106

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, binary_step=2, num_iter=200, GAMMA=0.5,
                 low_pass=100, clip_func=None, verbose=True, fast_victim=True, use_graph='auto'):
        self.fast_victim, self.use_graph = fast_victim, use_graph
        self._setup(model, adv_func, dist_func, attack_lr, binary_step, num_iter, GAMMA, clip_func, verbose,
                    low_pass=low_pass)

    def attack(self, data, target, y_truth=None):
        return self._run(data, target, y_truth)
