"""CW object-adding attack, interface of the reference's CW/Add_Objects.py (``CWAddObjects`` ctor :54-92,
``_init_centers`` :100-146, ``_rotate_shift`` :148-185, attack :187-367): ``num_add`` copies of a small object are
placed at salient cluster centres and their points, positions and y-rotation are optimised under ``L2ChamferDist``.
Best-result tracking and the bisection are device-resident."""
import copy

import numpy as np
import torch
import torch.optim as optim

from .Add import get_critical_points


def normalize_points_np(points):
    """util/pointnet_utils.py:107-113: centre and scale [K,3] to the unit ball."""
    points = points - np.mean(points, axis=0)[None, :]
    points = points / np.max(np.sqrt(np.sum(points ** 2, axis=1)), 0)
    assert np.sum(np.isnan(points)) == 0
    return points

from ._victim import Victim

class CWAddObjects:
    """Class for CW attack (adding objects)."""

    def __init__(self, model, adv_func, dist_func, object_pc, attack_lr=1e-2, init_weight=5., max_weight=40.,
                 binary_step=5, num_iter=500, num_add=3, obj_num_p=64, scaling=0.3, verbose=True, fast_victim=True):
        self.model = model.cuda()
        self.model.eval()
        self._victim = Victim(self.model, fast_victim)
        self.adv_func = adv_func
        self.dist_func = dist_func
        self.attack_lr = attack_lr
        self.init_weight = init_weight
        self.max_weight = max_weight
        self.binary_step = binary_step
        self.num_iter = num_iter
        self.num_add = num_add
        self.obj_num_p = obj_num_p
        self.verbose = verbose
        object_pc = self.process_object(object_pc, scaling)
        self.object_pc = np.zeros((self.num_add, self.obj_num_p, 3))
        for i in range(self.num_add):
            np.random.shuffle(object_pc)
            self.object_pc[i] = copy.deepcopy(object_pc[:self.obj_num_p])

    def process_object(self, pc, scaling):
        return normalize_points_np(pc) * scaling

    def _logits(self, x):
        return self._victim(x)

    def _init_centers(self, pc, label):
        """pc [B,3,K] -> np.ndarray [B,num_add,3]: the surface point closest to the mean of each of the largest DBSCAN
        clusters of the 128 critical points."""
        from sklearn.cluster import DBSCAN
        cri_points = get_critical_points(self.model, pc, label, 128)
        batch_cri = [[] for _ in range(len(pc))]
        for i in range(len(pc)):
            points = np.transpose(cri_points[i].detach().cpu().numpy(), [1, 0])
            result = DBSCAN(0.2, min_samples=3).fit_predict(points)
            keep = result > -0.5
            result, points = result[keep], points[keep]
            labels, counts = np.unique(result, return_counts=True)
            for one_label in labels[np.argsort(counts)[-self.num_add:]]:
                cluster_points = points[result == one_label]
                centre = np.mean(cluster_points, axis=0)
                batch_cri[i].append(copy.deepcopy(
                    cluster_points[np.argmin(np.sum((cluster_points - centre[None, :]) ** 2, axis=1))]))
            while len(batch_cri[i]) < self.num_add:
                batch_cri[i].append(copy.deepcopy(points[np.random.choice(len(points), 1)[0]]))
        return np.array(batch_cri)

    def _rotate_shift(self, points, angles, shifts):
        """points [B,num_add,obj_num_p,3], angles / shifts [B,num_add,3]: rotation about the y axis by angles[...,0]
        (the reference's own simplification, :158-159), then the shift."""
        batch = len(points)
        angle = angles[..., 0]
        c, s = torch.cos(angle), torch.sin(angle)
        zeros, ones = torch.zeros_like(c), torch.ones_like(c)
        rot = torch.stack([c, zeros, s, zeros, ones, zeros, -s, zeros, c], dim=-1).view(batch * self.num_add, 3, 3)
        rot_points = torch.bmm(points.view(batch * self.num_add, self.obj_num_p, 3), rot)
        return rot_points.view(batch, self.num_add, self.obj_num_p, 3) + shifts[:, :, None, :]

    def attack(self, data, target):
        """data [B,num_points,3], target [B] -> (o_bestdist float64 [B], float64 [B,num_points+num_add*obj_num_p,3], successes)."""
        self._victim.prepare()
        B, K = data.shape[:2]
        ori = data.float().cuda().detach().transpose(1, 2).contiguous()
        target = target.long().cuda().detach()
        dev = ori.device
        f64 = dict(device=dev, dtype=torch.float64)
        lower = torch.zeros(B, **f64)
        upper = torch.full((B,), float(self.max_weight), **f64)
        weight = torch.full((B,), float(self.init_weight), **f64)
        o_bestdist = torch.full((B,), 1e10, **f64)
        o_bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        n_add = self.num_add * self.obj_num_p
        o_bestattack = torch.zeros(B, 3, n_add, device=dev)
        shifts = torch.from_numpy(self._init_centers(ori, target)).float().cuda()
        objects = torch.from_numpy(np.tile(self.object_pc, (B, 1, 1, 1))).float().cuda()
        ori_t = ori.transpose(1, 2).contiguous()
        report_every = max(1, self.num_iter // 5)
        last_input = o_bestattack
        for binary_step in range(self.binary_step):
            adv_objects = (objects + torch.randn((B, self.num_add, self.obj_num_p, 3)).cuda() * 1e-7).requires_grad_()
            adv_shifts = (shifts + torch.randn((B, self.num_add, 3)).cuda() * 1e-7).requires_grad_()
            # the reference draws these with rand_like on a device tensor (:256-257, i.e. from the device generator); here
            # they come from the CPU generator like the two randn draws above, so a seeded run is device-independent
            adv_angles = (torch.rand((B, self.num_add, 3)).cuda() * np.pi).requires_grad_()
            bestdist = torch.full((B,), 1e10, **f64)
            bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
            opt = optim.Adam([adv_objects, adv_shifts, adv_angles], lr=self.attack_lr, weight_decay=0.)
            adv_loss = torch.zeros((), device=dev)
            dist_loss = torch.zeros((), device=dev)
            for iteration in range(self.num_iter):
                adv = self._rotate_shift(adv_objects, adv_angles, adv_shifts).view(B, n_add, 3)
                adv = adv.transpose(1, 2).contiguous()
                logits = self._logits(torch.cat([ori, adv], dim=-1))
                pred = logits.argmax(dim=-1)
                if self.verbose and iteration % report_every == 0:
                    print('Step {}, iteration {}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, iteration, (pred == target).sum().item(), B, adv_loss.item(), dist_loss.item()))
                adv_t = adv.transpose(1, 2).contiguous()
                with torch.no_grad():
                    last_input = adv.detach().clone()
                    dist_val = self.dist_func(adv_t, ori_t, adv_objects, objects, batch_avg=False).detach().double()
                    hit = pred == target
                    better = hit & (dist_val < bestdist)
                    bestdist = torch.where(better, dist_val, bestdist)
                    bestscore = torch.where(better, pred, bestscore)
                    o_better = hit & (dist_val < o_bestdist)
                    o_bestdist = torch.where(o_better, dist_val, o_bestdist)
                    o_bestscore = torch.where(o_better, pred, o_bestscore)
                    o_bestattack = torch.where(o_better[:, None, None], adv.detach(), o_bestattack)
                adv_loss = self.adv_func(logits, target).mean()
                dist_loss = self.dist_func(adv_t, ori_t, adv_objects, objects, weights=weight).mean()
                opt.zero_grad()
                (adv_loss + dist_loss).backward()
                opt.step()
                with torch.no_grad():
                    adv_angles.data = adv_angles.data % (2. * np.pi)
            with torch.no_grad():
                ok = (bestscore == target) & (bestscore != -1) & (bestdist <= o_bestdist)
                lower = torch.where(ok, torch.maximum(lower, weight), lower)
                upper = torch.where(ok, upper, torch.minimum(upper, weight))
                weight = (lower + upper) / 2.
        with torch.no_grad():
            best = torch.where((lower == 0.)[:, None, None], last_input, o_bestattack)
        success_num = int((lower > 0.).sum().item())
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        out = np.concatenate([ori.cpu().numpy().astype(np.float64), best.double().cpu().numpy()], axis=-1)
        return o_bestdist.cpu().numpy(), out.transpose((0, 2, 1)), success_num
