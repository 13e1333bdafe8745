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
import os as _os

# attack_many() runs independent attacks on their own HIP streams; with the runtime's default of 4 hardware queues four of
# them serialise again (bench.py: 23.9 vs 27.0 clouds/s).  Only effective if the HIP runtime has not started yet.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
