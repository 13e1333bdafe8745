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
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def farthest_point_sample(point, npoint):
    """point [N,D] -> the npoint rows picked by FPS on xyz, random start from ``np.random`` (as the reference)."""
    N = point.shape[0]
    xyz = point[:, :3]
    centroids = np.zeros((npoint,))
    distance = np.ones((N,)) * 1e10
    farthest = np.random.randint(0, N)
    for i in range(npoint):
        centroids[i] = farthest
        dist = np.sum((xyz - xyz[farthest, :]) ** 2, -1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = np.argmax(distance, -1)
    return point[centroids.astype(np.int32)]


class ModelNetDataLoader(Dataset):
    """``args`` needs ``num_point``, ``use_uniform_sample``, ``use_normals``, ``num_category`` (eval.py's parser)."""

    def __init__(self, root, args, split='train', process_data=False):
        self.root = root
        self.npoints = args.num_point
        self.process_data = process_data
        self.uniform = args.use_uniform_sample
        self.use_normals = args.use_normals
        self.num_category = args.num_category
        tag = 'modelnet10' if self.num_category == 10 else 'modelnet40'
        self.catfile = os.path.join(self.root, tag + '_shape_names.txt')
        self.cat = [line.rstrip() for line in open(self.catfile)]
        self.classes = dict(zip(self.cat, range(len(self.cat))))
        assert split in ('train', 'test')
        ids = [line.rstrip() for line in open(os.path.join(self.root, '%s_%s.txt' % (tag, split)))]
        names = ['_'.join(x.split('_')[0:-1]) for x in ids]
        self.datapath = [(names[i], os.path.join(self.root, names[i], ids[i]) + '.txt') for i in range(len(ids))]
        print('The size of %s data is %d' % (split, len(self.datapath)))
        self.save_path = os.path.join(root, 'modelnet%d_%s_%dpts%s.dat' % (self.num_category, split, self.npoints,
                                                                         '_fps' if self.uniform else ''))
        if self.process_data:
            if not os.path.exists(self.save_path):
                print('Processing data %s (only running in the first time)...' % self.save_path)
                self.list_of_points = [None] * len(self.datapath)
                self.list_of_labels = [None] * len(self.datapath)
                for index in range(len(self.datapath)):
                    self.list_of_points[index], self.list_of_labels[index] = self._read(index)
                with open(self.save_path, 'wb') as f:
                    pickle.dump([self.list_of_points, self.list_of_labels], f)
            else:
                print('Load processed data from %s...' % self.save_path)
                with open(self.save_path, 'rb') as f:
                    self.list_of_points, self.list_of_labels = pickle.load(f)

    def _read(self, index):
        name, path = self.datapath[index]
        label = np.array([self.classes[name]]).astype(np.int32)
        point_set = np.loadtxt(path, delimiter=',').astype(np.float32)
        if self.uniform:
            point_set = farthest_point_sample(point_set, self.npoints)
        else:
            point_set = point_set[0:self.npoints, :]
        return point_set, label

    def __len__(self):
        return len(self.datapath)

    def _get_item(self, index):
        if self.process_data:
            point_set, label = self.list_of_points[index], self.list_of_labels[index]
        else:
            point_set, label = self._read(index)
        point_set[:, 0:3] = pc_normalize(point_set[:, 0:3])
        if not self.use_normals:
            point_set = point_set[:, 0:3]
        return point_set, label[0]

    def __getitem__(self, index):
        return self._get_item(index)

    def test(self):
        self.npoints -= 1
