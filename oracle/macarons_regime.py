"""MACARONS-regime per-camera coverage gain (numpy restatement).  TEST INFRASTRUCTURE ONLY.

Restates macarons/utility/macarons_utils.py:
  Camera.get_points_in_fov :2400-2435 (given the camera matrices), get_distance_factor :1741-1765,
  get_distance_factor_threshold :1768-1776, get_distance_factor_smooth :1779-1788,
  predict_coverage_gain_for_single_camera :1580-1738 (both branches), Cell.fill :2551-2577.
PyTorch3D's transform_points is restated as p' = [x y z 1] M (row-vector), ndc = p'[:3] / p'[3]: the matrices are inputs.
Pinned (tests/test_oracle_golden.py) against goldens the reference's own functions produced on a real `Camera` object and
stand-in FoV cameras that supply exactly that transform (tests/golden/make_golden.py: gen_fov, gen_distance,
gen_single_camera, gen_cell).
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


def _focal_and_threshold(fov_deg, image_height, image_width, cell_resolution):
    """focal_length, pixel_size, epsilon, distance_th of macarons_utils.py:1752-1755 / :1780-1783 (fp32 tensor for the
    focal length, Python floats for the rest, like the reference)."""
    focal = F(1.0) / np.tan(F(np.pi / 180.) * F(fov_deg) / F(2.)).astype(F)
    pixel = 2. / min(image_height, image_width)
    eps = np.sqrt(np.pi) / 2. * cell_resolution
    return focal, pixel, eps, F(focal * F(eps) / F(pixel))


def distance_factor(pts, X_cam, fov_deg, image_height, image_width, cell_resolution):
    """get_distance_factor (:1741-1765): 1 inside distance_th, eps^2 (f / pixel / d)^2 beyond."""
    focal, pixel, eps, th = _focal_and_threshold(fov_deg, image_height, image_width, cell_resolution)
    d = np.linalg.norm(np.asarray(pts, F) - np.asarray(X_cam, F).reshape(1, 3), axis=-1, keepdims=True).astype(F)
    res = np.ones_like(d)
    far = d > th
    res[far] = (F(eps ** 2) * ((focal / F(pixel)) / d[far]) ** 2).astype(F)
    return res


def distance_factor_smooth(pts, X_cam, fov_deg, image_height, image_width, cell_resolution):
    """get_distance_factor_smooth (:1779-1788): 1 / (1 + (d / distance_th)^2)."""
    _, _, _, th = _focal_and_threshold(fov_deg, image_height, image_width, cell_resolution)
    d = np.linalg.norm(np.asarray(pts, F) - np.asarray(X_cam, F).reshape(1, 3), axis=-1, keepdims=True).astype(F)
    return (F(1.) / (F(1.) + (d / th) ** 2)).astype(F)


def cell_fill(cell_pts, pts, x_min, x_max, resolution, capacity, perm, n_point_min=0):
    """Cell.fill (:2551-2577) with the randperm draw given: strict bounding-box masks, fp64 admission test
    (dist to every point already in the cell > resolution), vstack, keep perm[:capacity].  Returns the new cell points."""
    pts = np.asarray(pts, F)
    m = (pts - np.asarray(x_max, F).reshape(1, 3)).max(axis=-1) < 0.
    add = pts[m]
    if add.shape[0] == 0:
        return cell_pts
    add = add[(add - np.asarray(x_min, F).reshape(1, 3)).min(axis=-1) > 0.]
    if add.shape[0] <= n_point_min:
        return cell_pts
    if cell_pts.shape[0] > 0:
        from . import scene
        add = add[scene.min_dist(add, cell_pts) > resolution]
    allp = np.vstack((cell_pts, add))
    return allp[np.asarray(perm)[:capacity]]


def coverage_gain_for_camera(sd_vis, X_world, vh, occ, rec, X_cam_world, M_pred, box_diag, samples, min_occ=0.1,
                             distance_th=17.0, dtype=np.float64, factor=None, return_parts=False):
    """factor: None -> get_distance_factor_threshold(distance_th); else a callable (pts_world [N,3], X_cam_world) -> [N,1]."""
    mask = points_in_fov(X_world, rec)
    occ_k = np.where(mask, np.asarray(occ, F).reshape(-1), F(0)).reshape(-1, 1)
    if not (occ_k[:, 0] > min_occ).any():
        return (0.0, None, None) if return_parts else 0.0
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
    fac = (distance_factor_threshold(world[:, :3], X_cam_world, distance_th) if factor is None else factor(world[:, :3], X_cam_world))[:, 0]
    gain = float((vis * fac).mean() * volume)
    return (gain, vis * fac, world) if return_parts else gain
