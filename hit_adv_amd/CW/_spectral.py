"""The graph-spectral tools of the AOF family (CW/AOF.py: ``knn`` :12-27, ``get_Laplace_from_pc`` :30-51): the 30-NN graph
comes from ``hitadv_knn_points`` (no [B,N,N] distance matrix, no full top-k), the dense Laplacian is assembled with one
scatter, and the eigendecomposition is ``torch.linalg.eigh`` (rocSOLVER) -- the reference calls ``torch.symeig`` (:50), which
no longer exists in torch >= 2."""
import torch

from ..pytorch3d_ops import knn_points


def knn(x, k):
    """x [B,3,N] -> idx [B,N,k]: the k nearest points (self included), nearest first."""
    pts = x.detach().transpose(2, 1).contiguous()
    return knn_points(pts, pts, K=k).idx


@torch.no_grad()
def get_Laplace_from_pc(ori_pc, k=30):
    """ori_pc [B,3,N] -> (eigenvalues [B,N] ascending, eigenvectors [B,N,N]) of L = D - A with
    A_ij = exp(-|x_i - x_j|^2) on the symmetrised k-NN graph."""
    pts = ori_pc.detach().transpose(2, 1).contiguous()
    nn = knn_points(pts, pts, K=k)
    B, N, _ = pts.shape
    A = torch.zeros(B, N, N, device=pts.device).scatter_(2, nn.idx, torch.exp(-nn.dists))
    A = torch.maximum(A, A.transpose(2, 1))  # (i,j) kept when either point is among the other's neighbours
    L = torch.diag_embed(A.sum(dim=2)) - A
    e, v = torch.linalg.eigh(L)
    return e.to(ori_pc), v.to(ori_pc)
