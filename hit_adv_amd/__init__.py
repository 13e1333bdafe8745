"""hit_adv_amd -- MI355X-native HiT-ADV hot path.

Module names mirror the reference tree so that its callers switch with an import prefix:
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.util.dist_utils import ChamferDist, HausdorffDist, KNNDist, ChamferkNNDist
    from hit_adv_amd.util.set_distance import chamfer, hausdorff
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    from hit_adv_amd.pytorch3d_ops import knn_points, knn_gather
    from hit_adv_amd.pointnet2_ops import pointnet2_utils
All compute runs in libhitadv_hip.so (hand-written HIP for gfx950); there is no CPU path.
"""
__version__ = "0.1.0"
