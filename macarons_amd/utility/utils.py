"""Host-side mirror of the two helpers of macarons/utility/utils.py that sit ON the hot path (the rest of that file --
config, mesh sampling, rasterisers, optimiser wrappers -- is out of scope):

  get_knn_points   utils.py:1497-1509   cdist + topk(largest=False) + knn_gather  ->  one HIP search (knn.hip)
  floor_divide     utils.py:113-117     Python-style (non-negative) modulo division used by the view-state binning

plus `knn_gather` with PyTorch3D's semantics (a pure index gather: plumbing), so that the names upstream's star-imports hand on
(`from .Attention import *`, Attention.py:5) resolve after `macarons_amd.patch_reference()`.
"""
import torch

from .. import ops


def get_knn_points(X, pc, k):
    """X [n_clouds, n_sample, 3], pc [n_clouds, seq_len, 3] -> (neighbours [n_clouds, n_sample, k, 3], distances [.., k] ascending,
    indices [.., k] int64).  Ties go to the lower index; d^2 = (dx^2 + dy^2) + dz^2 evaluated exactly."""
    return ops.knn_points(X, pc, k)


def floor_divide(x, d):
    """(x - x mod d) / d with torch's non-negative `%` -- the binning rule of compute_view_state (scone_utils.py:830-836)."""
    return (x - x % d) / d


def knn_gather(x, idx, lengths=None):
    """x [B,M,U], idx [B,L,K] -> [B,L,K,U]  (what pytorch3d.ops.knn_gather returns; no arithmetic)."""
    B, M, U = x.shape
    L, K = idx.shape[1], idx.shape[2]
    return torch.gather(x, 1, idx.reshape(B, L * K, 1).expand(-1, -1, U)).reshape(B, L, K, U)
