"""Dataset / checkpoint / log formats: the readers return, for the tiny tree of fixture g23, exactly
what the reference's Dataset/ModelNet.py and Dataset/ShapeNetDataLoader.py returned for it (same files, same numpy RNG
seed), and reference-style checkpoints load."""
import argparse
import json
import os

import numpy as np
import torch

from helpers import golden

HERE = os.path.dirname(os.path.abspath(__file__))


def _tree(root):
    files = json.load(open(os.path.join(HERE, 'golden', 'g23_dataset_tree.json')))
    for rel, text in files.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            f.write(text)


def test_modelnet_reader_matches_reference(tmp_path, capsys):
    from hit_adv_amd.Dataset.ModelNet import ModelNetDataLoader
    root = str(tmp_path)
    _tree(root)
    fx = golden('g23_datasets.npz')
    for tag, uniform, normals, process in (('plain', False, True, False), ('fps', True, False, False),
                                           ('cached', False, True, True), ('cached', False, True, True)):
        args = argparse.Namespace(num_point=16, use_uniform_sample=uniform, use_normals=normals, num_category=40)
        np.random.seed(11)
        ds = ModelNetDataLoader(root, args, split='test', process_data=process)
        assert len(ds) == 2
        items = [ds[i] for i in range(len(ds))]
        pts = np.stack([p for p, _ in items])
        assert pts.dtype == np.float32 and pts.shape == fx['modelnet_%s_points' % tag].shape
        np.testing.assert_array_equal(pts, fx['modelnet_%s_points' % tag])
        np.testing.assert_array_equal(np.array([l for _, l in items]), fx['modelnet_%s_labels' % tag])
    assert os.path.exists(os.path.join(root, 'modelnet40_test_16pts.dat'))  # second 'cached' pass read the pickle
    assert 'Load processed data' in capsys.readouterr().out


def test_shapenetpart_reader_matches_reference(tmp_path):
    from hit_adv_amd.Dataset.ShapeNetDataLoader import PartNormalDataset
    root = str(tmp_path)
    _tree(root)
    fx = golden('g23_datasets.npz')
    for tag, split, normals in (('test', 'test', True), ('trainval', 'trainval', False)):
        np.random.seed(13)
        ds = PartNormalDataset(root=root, npoints=12, split=split, normal_channel=normals)
        items = [ds[i] for i in range(len(ds))]
        np.testing.assert_array_equal(np.stack([p for p, _ in items]), fx['shapenet_%s_points' % tag])
        np.testing.assert_array_equal(np.array([l for _, l in items]), fx['shapenet_%s_labels' % tag])
    try:
        PartNormalDataset(root=root, split='nope')
        raise AssertionError('expected ValueError')
    except ValueError:
        pass


def test_reference_checkpoint_layout_loads(tmp_path):
    """eval.py:79,123: ``torch.load(path)['model_state_dict']`` with the reference's parameter names."""
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    from hit_adv_amd.util.other_utils import load_checkpoint
    torch.manual_seed(0)
    src = PointNetFeatureModel(40, normal_channel=False)
    path = os.path.join(str(tmp_path), 'PN_NT.checkpoint')
    torch.save({'epoch': 3, 'model_state_dict': src.state_dict()}, path)
    dst = PointNetFeatureModel(40, normal_channel=False)
    load_checkpoint(dst, path)
    for (ka, va), (kb, vb) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    torch.save({'state_dict': src.state_dict()}, path)  # save_checkpoint's layout (other_utils.py:173-184)
    load_checkpoint(PointNetFeatureModel(40, normal_channel=False), path)
    torch.save({'module.' + k: v for k, v in src.state_dict().items()}, path)  # bare DataParallel state_dict
    load_checkpoint(PointNetFeatureModel(40, normal_channel=False), path)


def test_logger_file_layout(tmp_path):
    from hit_adv_amd.util.other_utils import create_logger
    import logging
    logger = create_logger(str(tmp_path / 'log'), 'eval_last', 'info')
    logger.info('Overall attack success rate: 0.5000')
    for h in logging.getLogger().handlers:
        h.flush()
    text = open(str(tmp_path / 'log' / 'eval_last_log.txt')).read()
    assert text == 'Overall attack success rate: 0.5000\n'
    for h in list(logging.getLogger().handlers):
        if getattr(h, '_hitadv', False):
            logging.getLogger().removeHandler(h)
            h.close()


def test_synthetic_victim_helpers_are_pure_functions_of_their_arguments():
    """bench.py's victims and fixture g5d are described by (architecture, seed, gain, shake_bn's three numbers): the helpers must give
    the same model for the same numbers, touch only what they say they touch, and the surface-like clouds must be what the name says."""
    from hit_adv_amd.Dataset.synthetic import ToyVictim, shake_bn, sharpen, sphere_batch, synth_batch

    def victim():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Conv1d(3, 8, 1), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Conv1d(8, 4, 1),
                                   torch.nn.BatchNorm1d(4)).eval()
    a, b, c = shake_bn(victim(), seed=2), shake_bn(victim(), seed=2), shake_bn(victim(), seed=3)
    plain = victim()
    for (na, pa), (_, pb), (_, pc), (_, pp) in zip(a.state_dict().items(), b.state_dict().items(), c.state_dict().items(), plain.state_dict().items()):
        assert torch.equal(pa, pb)
        if 'running_mean' in na or 'running_var' in na:
            assert not torch.equal(pa, pp) and not torch.equal(pa, pc)
            if 'running_var' in na:
                assert float(pa.min()) >= 0.8 - 1e-6 and float(pa.max()) <= 1.2 + 1e-6
        else:
            assert torch.equal(pa, pp)  # weights, biases, affine parameters, counters untouched
    s = sharpen(victim(), 1.5)
    for (n, ps), (_, pp) in zip(s.state_dict().items(), plain.state_dict().items()):
        if n in ('0.weight', '3.weight'):
            assert torch.equal(ps, pp * 1.5)
        else:
            assert torch.equal(ps, pp)
    toy = ToyVictim()
    assert toy(torch.randn(2, 3, 50)).shape == (2, 40)
    d1, l1 = sphere_batch(3, 256, first=40)
    d2, _ = sphere_batch(3, 256, first=40)
    assert d1.shape == (3, 256, 6) and l1.shape[0] == 3 and torch.equal(d1, d2)
    r = d1[:, :, :3].norm(dim=2)
    assert float(r.min()) > 0.7 and float(r.max()) <= 1.0 + 1e-6 and float(r.std()) < 0.05  # a shell (centred, scaled to radius 1), not a ball
    g, _ = synth_batch(3, 256, first=40)
    assert float(g[:, :, :3].norm(dim=2).std()) > 3 * float(r.std())  # the Gaussian clouds fill their volume
    assert not torch.equal(sphere_batch(1, 256, first=41)[0], d1[:1])
