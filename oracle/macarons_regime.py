"""MACARONS-regime per-camera coverage gain (numpy restatement).  TEST INFRASTRUCTURE ONLY.

Restates macarons/utility/macarons_utils.py:
  Camera.get_points_in_fov :2400-2435 (given the camera matrices), get_distance_factor_threshold :1768-1776,
  predict_coverage_gain_for_single_camera :1580-1738 (forward-pass branch).
PyTorch3D's transform_points is restated as p' = [x y z 1] M (row-vector), ndc = p'[:3] / p'[3]; no reference test
pins results at that boundary (parity unpinned there), so the matrices are inputs.
"""
import numpy as np

from . import nets, scorer, view_state as V

F = np.float32


def points_in_fov(pts, rec):
    pts = np.asarray(pts, F)
    Mv, Mp = rec[:16].reshape(4, 4), rec[16:32].reshape(4, 4)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    lin = lambda M, j: ((x * M[0, j] + y * M[1, j]) + z * M[2, j]) + M[3, j]
    zv, px, py, pw = lin(Mv, 2), lin(Mp, 0), lin(Mp, 1), lin(Mp, 3)
    with np.errstate(divide="ignore", invalid="ignore"):
        nx, ny = px / pw, py / pw
    m = (nx >= rec[32]) & (nx <= rec[33]) & (ny >= rec[34]) & (ny <= rec[35]) & (zv > 0)
    if rec[39] > 0:
        d = pts - rec[36:39]
        m &= np.sqrt(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(F)) < rec[39]
    return m


def distance_factor_threshold(pts, X_cam, distance_th=17.0):
    d = np.linalg.norm(np.asarray(pts, F) - np.asarray(X_cam, F).reshape(1, 3), axis=-1, keepdims=True).astype(F)
    res = np.ones_like(d)
    far = d > distance_th
    res[far] = (F(distance_th) ** 2 / d[far] ** 2).astype(F)
    return res


def coverage_gain_for_camera(sd_vis, X_world, vh, occ, rec, X_cam_world, M_pred, box_diag, samples, min_occ=0.1,
                             distance_th=17.0, dtype=np.float64):
    mask = points_in_fov(X_world, rec)
    occ_k = np.where(mask, np.asarray(occ, F).reshape(-1), F(0)).reshape(-1, 1)
    if not (occ_k[:, 0] > min_occ).any():
        return 0.0
    res, res_h, inv, _ = V.sample_proxy_points(X_world, occ_k, vh, samples, min_occ, exact=True)
    volume = occ_k[occ_k[:, 0] > F(min_occ), 0].astype(np.float64).sum()
    world = res[inv]
    center_w = ((res[:, :3].max(0) + res[:, :3].min(0)) / 2).astype(F)
    tf = lambda p: (np.concatenate([p, np.ones((len(p), 1), F)], 1) @ M_pred)[:, :3].astype(F)
    center = tf(center_w[None])[0]
    pts = res.copy()
    pts[:, :3] = ((tf(res[:, :3]) - center) / F(box_diag)).astype(F)
    cam = ((tf(np.asarray(X_cam_world, F).reshape(1, 3)) - center) / F(box_diag)).astype(F)
    harm = nets.scone_vis_forward(sd_vis, pts[None], res_h[None], dtype)
    vis = scorer.compute_visibilities(pts[inv][None], harm[0][inv][None], cam[None], True, "trigfree", np.float64)[0, 0]
    fac = distance_factor_threshold(world[:, :3], X_cam_world, distance_th)[:, 0]
    return float((vis * fac).mean() * volume)
