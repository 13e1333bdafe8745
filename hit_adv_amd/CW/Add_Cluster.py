"""CW cluster-adding attack, interface of the reference's CW/Add_Cluster.py (``CWAddClusters`` ctor :52-81,
``_init_centers`` :83-130, attack :132-278): ``num_add`` clusters of ``cl_num_p`` points, initialised on DBSCAN
clusters of the 128 most salient points, optimised under ``FarChamferDist``.  The optimisation loop is CWAdd's; only
the initialisation differs (host-side scikit-learn DBSCAN and ``np.random.choice`` draws, as in the reference)."""
import copy

import numpy as np
import torch

from .Add import CWAdd, get_critical_points


class CWAddClusters(CWAdd):
    """Class for CW attack (adding clusters)."""

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, init_weight=5., max_weight=30., binary_step=5,
                 num_iter=500, num_add=3, cl_num_p=32, verbose=True):
        super().__init__(model, adv_func, dist_func, attack_lr=attack_lr, init_weight=init_weight,
                         max_weight=max_weight, binary_step=binary_step, num_iter=num_iter, num_add=num_add,
                         verbose=verbose)
        self.cl_num_p = cl_num_p

    def _init_centers(self, pc, label):
        """pc [B,3,K] -> np.ndarray [B,num_add,cl_num_p,3]: the largest DBSCAN clusters of the critical points."""
        from sklearn.cluster import DBSCAN
        cri_points = get_critical_points(self.model, pc, label, 128)
        batch_cri = [[] for _ in range(len(pc))]
        for i in range(len(pc)):
            points = np.transpose(cri_points[i].detach().cpu().numpy(), [1, 0])  # [128,3]
            result = DBSCAN(0.2, min_samples=3).fit_predict(points)
            keep = result > -0.5
            result, points = result[keep], points[keep]
            labels, counts = np.unique(result, return_counts=True)
            for one_label in labels[np.argsort(counts)[-self.num_add:]]:
                cluster_points = points[result == one_label]
                replace = not (len(cluster_points) > self.cl_num_p)
                sel = np.random.choice(len(cluster_points), self.cl_num_p, replace=replace)
                batch_cri[i].append(copy.deepcopy(cluster_points[sel]))
            while len(batch_cri[i]) < self.num_add:  # not enough clusters: the neighbourhood of a random critical point
                rand_point = points[np.random.choice(len(points), 1)[0]]
                order = np.argsort(np.sum((points - rand_point[None, :]) ** 2, axis=1))[:self.cl_num_p]
                batch_cri[i].append(copy.deepcopy(points[order]))
        return np.array(batch_cri)

    def _init_points(self, ori, target):
        clusters = torch.from_numpy(self._init_centers(ori, target)).float().cuda()
        B = ori.shape[0]
        return clusters.view(B, self.num_add * self.cl_num_p, 3).transpose(1, 2).contiguous()
