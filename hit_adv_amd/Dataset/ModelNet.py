"""ModelNet40 / ModelNet10 "normal_resampled" reader, interface of the reference's Dataset/ModelNet.py
(``pc_normalize`` :12-17, ``farthest_point_sample`` :20-41, ``ModelNetDataLoader`` :44-135).

On-disk format (unchanged): ``<root>/modelnet40_shape_names.txt`` (one class per line, its index is the label),
``<root>/modelnet40_{train,test}.txt`` (one shape id per line, e.g. ``airplane_0627``), and per shape
``<root>/<class>/<shape id>.txt`` with one point per line, comma separated ``x,y,z,nx,ny,nz`` (10,000 points).
``process_data=True`` caches the first ``num_point`` points (or an FPS subsample) of every shape in a pickle
``modelnet40_<split>_<npts>pts[_fps].dat`` = ``[list_of_points, list_of_labels]`` -- files written by the reference are
read as they are.  Items are ``(float32 [num_point, 3 or 6], int32 label)``; xyz is centred and scaled to the unit ball.

Host-side code (it runs inside DataLoader workers); the GPU path starts at ``eval_ASR`` / ``attack``.
"""
import os
import pickle

import numpy as np
from torch.utils.data import Dataset


def pc_normalize(pc):
    """Centre on the mean, scale the farthest point onto the unit sphere."""
    centred = pc - pc.mean(axis=0)
    return centred / np.sqrt((centred ** 2).sum(axis=1)).max()


def farthest_point_sample(point, npoint):
    """point [N,D] -> the npoint rows picked by farthest point sampling on xyz.  One ``np.random.randint`` draw for the
    start, first arg-max on ties: the selection (and the RNG stream) of the reference's loop."""
    xyz = point[:, :3]
    chosen = np.empty(npoint, dtype=np.int64)
    nearest = np.full(xyz.shape[0], 1e10)
    cur = np.random.randint(0, xyz.shape[0])
    for i in range(npoint):
        chosen[i] = cur
        np.minimum(nearest, ((xyz - xyz[cur]) ** 2).sum(-1), out=nearest)
        cur = int(nearest.argmax())
    return point[chosen]


class ModelNetDataLoader(Dataset):
    """``args`` needs ``num_point``, ``use_uniform_sample``, ``use_normals``, ``num_category`` (eval.py's parser)."""

    def __init__(self, root, args, split='train', process_data=False):
        if split not in ('train', 'test'):
            raise AssertionError(split)
        self.root, self.process_data = root, process_data
        self.npoints, self.uniform = args.num_point, args.use_uniform_sample
        self.use_normals, self.num_category = args.use_normals, args.num_category
        family = 'modelnet%d' % (10 if self.num_category == 10 else 40)
        self.catfile = os.path.join(root, family + '_shape_names.txt')
        self.cat = self._lines(self.catfile)
        self.classes = {name: i for i, name in enumerate(self.cat)}
        self.datapath = []
        for shape_id in self._lines(os.path.join(root, '%s_%s.txt' % (family, split))):
            cls = shape_id.rsplit('_', 1)[0]  # 'night_stand_0001' -> 'night_stand'
            self.datapath.append((cls, os.path.join(root, cls, shape_id) + '.txt'))
        print('The size of %s data is %d' % (split, len(self.datapath)))
        suffix = '_fps' if self.uniform else ''
        self.save_path = os.path.join(root, 'modelnet%d_%s_%dpts%s.dat' % (self.num_category, split, self.npoints, suffix))
        if process_data:
            self.list_of_points, self.list_of_labels = self._cached()

    @staticmethod
    def _lines(path):
        with open(path) as f:
            return [line.rstrip() for line in f]

    def _cached(self):
        """The reference's pickle cache: ``[list_of_points, list_of_labels]`` in ``self.save_path``."""
        if os.path.exists(self.save_path):
            print('Load processed data from %s...' % self.save_path)
            with open(self.save_path, 'rb') as f:
                return pickle.load(f)
        print('Processing data %s (only running in the first time)...' % self.save_path)
        pairs = [self._read(i) for i in range(len(self.datapath))]
        points, labels = [p for p, _ in pairs], [l for _, l in pairs]
        with open(self.save_path, 'wb') as f:
            pickle.dump([points, labels], f)
        return points, labels

    def _read(self, index):
        cls, path = self.datapath[index]
        cloud = np.loadtxt(path, delimiter=',').astype(np.float32)
        cloud = farthest_point_sample(cloud, self.npoints) if self.uniform else cloud[:self.npoints, :]
        return cloud, np.array([self.classes[cls]]).astype(np.int32)

    def __len__(self):
        return len(self.datapath)

    def _get_item(self, index):
        if self.process_data:
            cloud, label = self.list_of_points[index], self.list_of_labels[index]
        else:
            cloud, label = self._read(index)
        cloud[:, 0:3] = pc_normalize(cloud[:, 0:3])  # in place, as the reference: a cached cloud is normalised again
        return (cloud if self.use_normals else cloud[:, 0:3]), label[0]

    def __getitem__(self, index):
        return self._get_item(index)

    def test(self):
        self.npoints -= 1
