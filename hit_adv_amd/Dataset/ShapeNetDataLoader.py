"""ShapeNetPart reader, interface of the reference's Dataset/ShapeNetDataLoader.py (``PartNormalDataset`` :137-235).

On-disk format (unchanged): ``<root>/synsetoffset2category.txt`` (``<category> <synset dir>`` per line; the line index
is the class label), ``<root>/train_test_split/shuffled_{train,val,test}_file_list.json`` (lists of
``shape_data/<synset>/<token>``), and per shape ``<root>/<synset>/<token>.txt`` with whitespace separated
``x y z nx ny nz seg``.  Items are ``(float32 [npoints, 3 or 6], int32 class)``: xyz normalised to the unit ball, then
``npoints`` rows drawn WITH replacement from ``np.random.choice`` (the segmentation label is read and dropped, as in the
reference's classification use, eval.py:92-104)."""
import json
import os

import numpy as np
from torch.utils.data import Dataset


def pc_normalize(pc):
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


class PartNormalDataset(Dataset):
    seg_classes = {'Earphone': [16, 17, 18], 'Motorbike': [30, 31, 32, 33, 34, 35], 'Rocket': [41, 42, 43],
                   'Car': [8, 9, 10, 11], 'Laptop': [28, 29], 'Cap': [6, 7], 'Skateboard': [44, 45, 46],
                   'Mug': [36, 37], 'Guitar': [19, 20, 21], 'Bag': [4, 5], 'Lamp': [24, 25, 26, 27],
                   'Table': [47, 48, 49], 'Airplane': [0, 1, 2, 3], 'Pistol': [38, 39, 40],
                   'Chair': [12, 13, 14, 15], 'Knife': [22, 23]}

    def __init__(self, root='./data/shapenetcore_partanno_segmentation_benchmark_v0_normal', npoints=2500,
                 split='train', class_choice=None, normal_channel=False):
        self.npoints = npoints
        self.root = root
        self.normal_channel = normal_channel
        self.cat = {}
        with open(os.path.join(self.root, 'synsetoffset2category.txt'), 'r') as f:
            for line in f:
                ls = line.strip().split()
                self.cat[ls[0]] = ls[1]
        self.classes_original = dict(zip(self.cat, range(len(self.cat))))
        if class_choice is not None:
            self.cat = {k: v for k, v in self.cat.items() if k in class_choice}

        def ids(name):
            with open(os.path.join(self.root, 'train_test_split', 'shuffled_%s_file_list.json' % name), 'r') as f:
                return set(str(d.split('/')[2]) for d in json.load(f))
        train_ids, val_ids, test_ids = ids('train'), ids('val'), ids('test')
        keep = {'trainval': train_ids | val_ids, 'train': train_ids, 'val': val_ids, 'test': test_ids}
        if split not in keep:
            raise ValueError('Unknown split: %s' % split)  # the reference prints and calls exit(-1)
        self.meta = {}
        for item in self.cat:
            dir_point = os.path.join(self.root, self.cat[item])
            fns = [fn for fn in sorted(os.listdir(dir_point)) if fn[0:-4] in keep[split]]
            self.meta[item] = [os.path.join(dir_point, os.path.splitext(os.path.basename(fn))[0] + '.txt') for fn in fns]
        self.datapath = [(item, fn) for item in self.cat for fn in self.meta[item]]
        self.classes = {i: self.classes_original[i] for i in self.cat.keys()}
        self.cache = {}
        self.cache_size = 20000

    def __getitem__(self, index):
        if index in self.cache:
            point_set, cls, seg = self.cache[index]
        else:
            cat, path = self.datapath[index]
            cls = np.array([self.classes[cat]]).astype(np.int32)
            data = np.loadtxt(path).astype(np.float32)
            point_set = data[:, 0:6] if self.normal_channel else data[:, 0:3]
            seg = data[:, -1].astype(np.int32)
            if len(self.cache) < self.cache_size:
                self.cache[index] = (point_set, cls, seg)
        point_set[:, 0:3] = pc_normalize(point_set[:, 0:3])
        choice = np.random.choice(len(seg), self.npoints, replace=True)
        return point_set[choice, :], cls[0]

    def __len__(self):
        return len(self.datapath)
