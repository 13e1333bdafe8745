"""Untargeted kNN attack, interface of the reference's CW/UKNN.py::CWUKNN (:14-159).

Differences from CWKNN, all taken from the reference: the constructor ends with ``pre_head=None`` (:18-19) and that
module, when given, is applied to the adversarial cloud in front of every victim forward (:82-85, :141-144);
``clip_func`` also receives the normals (:120-122); success -- in the progress lines and in the returned count --
means ``pred != target`` (:95, :149).  The loop itself is CWKNN's (fixed buffers, one iteration replayed as a
hipGraph when nothing in it needs the host)."""
from .kNN import CWKNN


class CWUKNN(CWKNN):
    """Class for CW attack."""

    def __init__(self, model, adv_func, dist_func, clip_func, attack_lr=1e-3, num_iter=2500, pre_head=None,
                 verbose=True, fast_victim=True, use_graph='auto'):
        super().__init__(model, adv_func, dist_func, clip_func, attack_lr=attack_lr, num_iter=num_iter,
                         verbose=verbose, fast_victim=fast_victim, use_graph=use_graph)
        self.pre_head = pre_head

    def _logits(self, x):
        return self._victim(self.pre_head(x) if self.pre_head is not None else x)

    def _clip(self, adv, ori, normal):
        return self.clip_func(adv, ori, normal)

    @staticmethod
    def _success(pred, target):
        return pred != target
