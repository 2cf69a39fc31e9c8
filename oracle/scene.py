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


# ---- compute_scene_occupancy_probability_field (macarons_utils.py:1395-1540) on plain arrays ---------------------------------
def occupancy_field(sd_occ, x_min, x_max, grid, surface_cells, proxy_cells, proxy, sup_occ, out_of_field, view_states, proba,
                    Mpred, perms, k_for_knn=16, neighborhood=3, chunk=20000, dtype=np.float32):
    """surface_cells / proxy_cells: {(i,j,k): points [n,3]} / {(i,j,k): proxy indices in storage order}; perms: the randperm
    draws in the order the reference makes them (3 per chunk per processed cell).  Returns (X_world, view_harmonics, occ_probs,
    proba_after)."""
    from . import nets, view_state as V
    F = np.float32
    x_min, x_max, proxy = np.asarray(x_min, F), np.asarray(x_max, F), np.asarray(proxy, F)
    grid = [int(g) for g in grid]
    step = ((x_max - x_min) / np.array(grid, F)).astype(F)
    proba = np.asarray(proba, F).copy()
    occ_mask = np.asarray(sup_occ).reshape(-1) > 0
    fov_mask = np.asarray(out_of_field).reshape(-1) < 1
    seen = occ_mask & fov_mask
    proba[seen] = 0
    d = proxy[seen] - x_min
    idx = np.minimum(((d - np.mod(d, step)) / step).astype(np.int64), np.array(grid) - 1).clip(min=0)
    cells = np.unique(idx, axis=0)                                   # torch.unique(dim=0): lexicographic order
    base, hp, ha = V.all_harmonics_under_degree(8, 7, 14)
    Mv = np.asarray(Mpred, F).reshape(4, 4)
    tf = lambda p: (((p[:, 0:1] * Mv[0, :3] + p[:, 1:2] * Mv[1, :3]) + p[:, 2:3] * Mv[2, :3]) + Mv[3, :3]).astype(F)
    grid_dirs = (V.view_space_grid(7, 14) @ Mv[:3, :3].T).astype(F)
    Xs, Hs, Os, pi = [], [], [], 0
    for c in cells:
        ci, cj, ck = [int(v) for v in c]
        neigh = sorted({(min(max(ci + a, 0), grid[0] - 1), min(max(cj + b, 0), grid[1] - 1), min(max(ck + e, 0), grid[2] - 1))
                        for a in (-1, 0, 1) for b in (-1, 0, 1) for e in (-1, 0, 1)})
        pcw = np.concatenate([surface_cells[n] for n in neigh if n in surface_cells] or [np.zeros((0, 3), F)])
        members = np.zeros(len(proxy), bool)
        members[np.asarray(proxy_cells.get((ci, cj, ck), []), np.int64)] = True
        cmask = members & occ_mask
        Xw = proxy[cmask]
        if not (len(pcw) > 4 * k_for_knn and len(Xw) > 0):
            continue
        center_w = (x_min + (np.array([ci, cj, ck], F) + F(0.5)) * step).astype(F)
        center = tf(center_w[None])[0]
        diag = F(neighborhood) * np.linalg.norm((step).astype(F)).astype(F)
        pc = ((tf(pcw) - center) / diag).astype(F)[None]
        X = ((tf(Xw) - center) / diag).astype(F)[None]
        vs, _ = V.move_view_state_to_view_space(np.asarray(view_states, F)[cmask][None], grid_dirs, 7, 14)
        vh = V.compute_view_harmonics(vs, base, hp, ha, 7, 14)
        occ = []
        for lo in range(0, X.shape[1], chunk):
            M = pc.shape[1]
            ds = int(np.power(M / (16 * 8), 1. / 2)) or 2
            p3 = [perms[pi][:2048], perms[pi + 1][:M // ds], perms[pi + 2][:(M // ds) // ds]]
            pi += 3
            occ.append(nets.scone_occ_forward(sd_occ, pc, X[:, lo:lo + chunk], vh[:, lo:lo + chunk], p3, dtype).reshape(-1, 1))
        occ = np.concatenate(occ).astype(F)
        Xs.append(Xw); Hs.append(vh[0]); Os.append(occ)
        proba[cmask] = occ[:, 0]
    oof = np.asarray(out_of_field).reshape(-1) > 0
    Xs.append(proxy[oof]); Hs.append(np.zeros((int(oof.sum()), 64), F)); Os.append(proba[oof][:, None])
    return np.concatenate(Xs), np.concatenate(Hs), np.concatenate(Os), proba
