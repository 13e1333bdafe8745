"""AOF attack ("Boosting 3D adversarial attacks with attacking on frequency"), interface of the reference's
CW/AOF.py (``knn`` :12-27, ``get_Laplace_from_pc`` :30-51, ``CWAOF`` :54-241): Adam on the low-frequency
component of the cloud in the eigenbasis of its kNN-graph Laplacian, GAMMA-weighted loss on the full cloud
and on its low-frequency part.

GPU plan: the graph-spectral tools live in ``CW/_spectral.py`` (HIP kNN, one scatter, ``torch.linalg.eigh``); the
attack loop is the shared engine of ``CW/_family.py`` -- device-resident best tracking, one captured iteration replayed.
"""
from ._family import _CWFamily
from ._spectral import get_Laplace_from_pc, knn  # noqa: F401  (the reference's module exports both)


class CWAOF(_CWFamily):
    """Class for the AOF attack (constructor of CW/AOF.py:58-81).  The loop is the shared engine of CW/_family.py with the
    spectral switch on: untargeted, no auto-encoder view, loss weights (1 - GAMMA, GAMMA), predictions re-evaluated after the
    clip (:121-128), final clip of the result (:137); captured into a hipGraph when the victim allows it."""
    spectral = True
    targeted = False
    fresh = True

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, binary_step=2, num_iter=200, GAMMA=0.5,
                 low_pass=100, clip_func=None, verbose=True, fast_victim=True, use_graph='auto'):
        self.fast_victim, self.use_graph = fast_victim, use_graph
        self._setup(model, adv_func, dist_func, attack_lr, binary_step, num_iter, GAMMA, clip_func, verbose, low_pass=low_pass)

    def attack(self, data, target):
        """data [B,num_points,3|6], target [B] (true labels; untargeted) -> (float32 ndarray [B,num_points,3], successes)."""
        return self._run(data, target)[1:]

    @staticmethod
    def _shape_result(out):
        return out[1:]
