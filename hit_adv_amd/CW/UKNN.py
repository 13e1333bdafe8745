"""Untargeted kNN attack, interface of the reference's CW/UKNN.py::CWUKNN (:41-159): identical to
CWKNN except that ``clip_func`` also receives the normals (:120-122) and success means
``pred != target`` (:87,153)."""
from .kNN import CWKNN


class CWUKNN(CWKNN):

    def _clip(self, adv, ori, normal):
        return self.clip_func(adv, ori, normal)

    def attack(self, data, target):
        adv, hit = super().attack(data, target)
        return adv, data.shape[0] - hit
