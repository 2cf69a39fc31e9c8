"""Scene-side point bookkeeping (numpy restatement).  TEST INFRASTRUCTURE ONLY.
  min_dist: torch.min(torch.cdist(a.double(), b.double()), dim=-1)  (macarons_utils.py:2566, 3022, 3049)
  unproject_depth: Camera.project_depth_in_3D :2339-2360 + pytorch3d FoVPerspectiveCameras.unproject_points
  (pytorch3d 0.6.2, not vendored: its published algorithm restated);
  project_depth_back_to_3D: utils.py:1458-1487; compute_partial_point_cloud: macarons_utils.py:2362-2398.
Pinned (tests/test_oracle_golden.py) against the reference's functions run on stand-in cameras (make_golden.py: gen_unproject,
gen_cell)."""
import numpy as np


def min_dist(A, B):
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    if len(B) == 0:
        return np.full(len(A), np.inf)
    d2 = ((A[:, None, :] - B[None, :, :]) ** 2)
    return np.sqrt(((d2[..., 0] + d2[..., 1]) + d2[..., 2]).min(axis=1))


def ndc_tabs(H, W):
    m = np.float32(min(H, W))
    i = np.arange(H, dtype=np.float32)[:, None].repeat(W, 1)
    j = np.arange(W, dtype=np.float32)[None, :].repeat(H, 0)
    return (np.float32(W) / m - (j / (m - 1)) * 2).astype(np.float32), (np.float32(H) / m - (i / (m - 1)) * 2).astype(np.float32)


def unproject_depth(depth, Minv, k22, k32):
    H, W = depth.shape
    nx, ny = ndc_tabs(H, W)
    d = depth.astype(np.float32)
    sd = ((np.float32(k22) * d + np.float32(k32)) / d).astype(np.float32)
    p = np.stack([nx, ny, sd, np.ones_like(sd)], -1).reshape(-1, 4).astype(np.float64) @ np.asarray(Minv, np.float64)
    return (p[:, :3] / p[:, 3:4]).astype(np.float32)


def project_depth_back_to_3D(depth, Minv, k22, k32):
    """utils.py:1458-1487: depth [n,H,W,1] -> world points of the pixels with depth > -1, camera-major."""
    out = []
    for c in range(depth.shape[0]):
        d = np.asarray(depth[c, ..., 0], np.float32)
        w = unproject_depth(d, Minv[c], k22[c], k32[c])
        out.append(w[(d > -1).reshape(-1)])
    return np.concatenate(out, 0)


def compute_partial_point_cloud(depth, mask, Minv, k22, k32, gathering_factor, fov_range, perm):
    """macarons_utils.py:2362-2398 with the randperm draw given."""
    d = np.asarray(depth[0, ..., 0], np.float32)
    keep = np.asarray(mask).reshape(-1).astype(bool)
    if fov_range is not None:
        keep &= (d < fov_range).reshape(-1)
    world = unproject_depth(d, Minv, k22, k32)[keep]
    n = int(len(world) * gathering_factor)
    return world[np.asarray(perm)[:n]]
