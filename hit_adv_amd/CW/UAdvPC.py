"""Untargeted AdvPC attack, interface of the reference's CW/UAdvPC.py::CWUAdvPC (ctor :14-38, attack :40-167)."""
from ._family import _CWFamily


class CWUAdvPC(_CWFamily):
    """Class for CW UAdvPC attack."""
    fresh = False  # best-tracking uses the predictions of the logits the loss was computed on (:109-111)

    def __init__(self, model, ae_model, adv_func, dist_func, attack_lr=1e-2, binary_step=2, num_iter=200, GAMMA=0.5,
                 clip_func=None, verbose=True, fast_victim=True, use_graph='auto'):
        self.fast_victim, self.use_graph = fast_victim, use_graph
        self._setup(model, adv_func, dist_func, attack_lr, binary_step, num_iter, GAMMA, clip_func, verbose,
                    ae_model=ae_model)

    def attack(self, data, target):
        """data [B,num_points,3], target [B] (true labels) -> (o_bestdist float64 [B], float32 [B,num_points,3], successes)."""
        return self._run(data, target)
