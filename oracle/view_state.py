"""View-state binning, view harmonics and occupancy-weighted proxy sampling (numpy restatement).
TEST INFRASTRUCTURE ONLY.

Restates (upstream tree):
  macarons/utility/utils.py:113-117            floor_divide  (Python-style, non-negative mod)
  macarons/utility/scone_utils.py:714-738      get_all_harmonics_under_degree
  macarons/utility/scone_utils.py:799-860      compute_view_state
  macarons/utility/scone_utils.py:934-960      compute_view_harmonics
  macarons/utility/scone_utils.py:1030-1076    sample_proxy_points
"""
import numpy as np

from . import sh

F = np.float32


def floor_divide(x, d):
    return ((x - np.mod(x, d)) / d).astype(x.dtype)          # utils.py:116


def all_harmonics_under_degree(degree, n_elev, n_azim, dtype=np.float32):
    """-> (z [degree^2, n_elev*n_azim], h_polar, h_azim); elevation-major grid (scone_utils.py:723-727)."""
    assert degree == sh.MAX_RANK
    h_elev = np.array([-np.pi / 2 + (i + 1) / (n_elev + 1) * np.pi for i in range(n_elev) for j in range(n_azim)], dtype=F)
    h_polar = (-h_elev + F(np.pi / 2)).astype(F)
    h_azim = np.array([2 * np.pi * j / n_azim for i in range(n_elev) for j in range(n_azim)], dtype=F)
    z = sh.sh_basis_literal(h_polar.astype(dtype), h_azim.astype(dtype), dtype)
    return np.ascontiguousarray(z.T), h_polar, h_azim


def view_state_indices(pts, X_view, n_elev, n_azim):
    """Flat bin index of every (point, view) pair, literally as scone_utils.py:815-849 in fp32."""
    pts = np.asarray(pts, F)
    X_view = np.asarray(X_view, F)
    n_clouds, seq_len, _ = pts.shape
    n_view = X_view.shape[0]
    elev_step, azim_step = F(np.pi / (n_elev + 1)), F(2 * np.pi / n_azim)
    rays = (X_view[None, None, :, :] - pts[:, :, None, :3]).reshape(-1, 3)
    _, ray_elev, ray_azim = sh.spherical_coords(rays)
    idx_elev = floor_divide(ray_elev, elev_step)
    idx_azim = floor_divide(ray_azim, azim_step)
    idx_elev[np.mod(ray_elev, elev_step) > F(np.pi / (n_elev + 1) / 2.)] += 1
    idx_azim[np.mod(ray_azim, azim_step) > F(2 * np.pi / n_azim / 2.)] += 1
    idx_elev[idx_elev >= n_elev] = n_elev - 1
    idx_elev[idx_elev < -n_elev // 2] = -n_elev // 2            # Python precedence: (-7)//2 = -4   (:839)
    idx_azim[idx_azim > n_azim // 2] = -n_azim // 2             # (-14)//2 = -7                    (:842)
    idx_elev += n_elev // 2
    idx_azim[idx_azim < 0] += n_azim
    indices = idx_elev.astype(np.int64) * n_azim + idx_azim.astype(np.int64)
    indices %= n_elev * n_azim
    return indices.reshape(n_clouds, seq_len, n_view)


def compute_view_state(pts, X_view, n_elev, n_azim):
    idx = view_state_indices(pts, X_view, n_elev, n_azim)
    B, Q, V = idx.shape
    vs = np.zeros((B, Q, n_elev * n_azim), F)
    b, q, _ = np.meshgrid(np.arange(B), np.arange(Q), np.arange(V), indexing="ij")
    vs[b, q, idx] = 1.0                                          # idempotent write (:857-858)
    return vs


def bin_boundary_margin(pts, X_view, n_elev, n_azim):
    """fp64 distance (radians) of every (point, view) ray to the nearest binning decision boundary; pairs with a
    tiny margin are legitimately ambiguous between implementations of asin/acos (tie-aware parity checks)."""
    pts = np.asarray(pts, np.float64)
    X_view = np.asarray(X_view, np.float64)
    rays = (X_view[None, None, :, :] - pts[:, :, None, :3]).reshape(-1, 3)
    r = np.linalg.norm(rays, axis=1)
    elev = np.arcsin(np.clip(rays[:, 1] / r, -1, 1))
    azim = np.arctan2(rays[:, 0], rays[:, 2])
    es, as_ = np.pi / (n_elev + 1), 2 * np.pi / n_azim
    de = np.abs(np.mod(elev, es) - es / 2)
    da = np.abs(np.mod(azim, as_) - as_ / 2)
    da = np.minimum(da, np.pi - np.abs(azim))                     # the +-pi seam
    return np.minimum(de, da).reshape(pts.shape[0], pts.shape[1], X_view.shape[0])


def view_harmonics_matrix(base_harmonics, h_polar, n_elev, n_azim):
    """The constant [64, 98] matrix of scone_utils.py:958: base * sin(polar) * polar_step * azim_step (fp32 chain)."""
    polar_step, azim_step = np.pi / (n_elev + 1), 2 * np.pi / n_azim
    m = np.asarray(base_harmonics, F) * np.sin(np.asarray(h_polar, F))[None, :]
    m = (m * F(polar_step)).astype(F)
    return (m * F(azim_step)).astype(F)


def compute_view_harmonics(view_state, base_harmonics, h_polar, h_azim, n_elev, n_azim):
    m = view_harmonics_matrix(base_harmonics, h_polar, n_elev, n_azim)
    return (np.asarray(view_state, F)[..., None, :] * m).sum(axis=-1, dtype=F).astype(F)


def sample_proxy_points(X_world, preds, view_harmonics, samples, min_occ, exact=False):
    """scone_utils.py:1030-1061 with the uniforms `samples` [n_sample] given explicitly.
    exact=False: fp32 sequential cumsum / compare like the reference on CPU;
    exact=True : fp64 CDF, first index with C_i >= u * S  (the convention the HIP kernel implements).
    Returns (res [n_u,4], res_harmonics [n_u,64], inverse_idx [n_sample], unique original indices [n_u])."""
    X_world, preds, vh = np.asarray(X_world, F), np.asarray(preds, F), np.asarray(view_harmonics, F)
    samples = np.asarray(samples, F).reshape(-1)
    mask = preds[..., 0] > F(min_occ)
    orig = np.nonzero(mask)[0]
    res_X, res_preds, res_h = X_world[mask], preds[mask], vh[mask]
    if exact:
        c = np.cumsum(res_preds[:, 0].astype(np.float64))
        target = samples.astype(np.float64) * c[-1]
        res_idx = np.minimum(np.searchsorted(c, target, side="left"), len(c) - 1)
    else:
        p = (res_preds[:, 0] / res_preds.sum(dtype=F)).astype(F)
        c = np.cumsum(p, dtype=F)
        d = c[None, :] - samples[:, None]
        d[d < 0] = 2
        res_idx = np.argmin(d, axis=-1)
    uniq, inverse = np.unique(res_idx, return_inverse=True)
    res = np.concatenate((res_X[uniq], res_preds[uniq]), axis=-1)
    return res, res_h[uniq], inverse.astype(np.int64), orig[uniq]


def sampler_tie_aware_match(res_a, inv_a, res_b, inv_b, X_world, preds, min_occ, max_mismatch=4):
    """Two sampling results agree if, sample by sample, they picked the same point, except for at most a few
    samples whose uniform falls within rounding distance of a CDF step (fp32 sequential cumsum of the reference
    vs the exact CDF): those must have picked ADJACENT kept points."""
    pa, pb = np.asarray(res_a)[np.asarray(inv_a)], np.asarray(res_b)[np.asarray(inv_b)]      # per-sample (x,y,z,occ)
    if pa.shape != pb.shape:
        return False
    diff = np.nonzero((pa != pb).any(axis=1))[0]
    if len(diff) > max_mismatch:
        return False
    kept = np.nonzero(np.asarray(preds)[..., 0] > np.float32(min_occ))[0]
    Xk = np.asarray(X_world, np.float32)[kept]
    for s in diff:
        ia = np.nonzero((Xk == pa[s, :3]).all(axis=1))[0]
        ib = np.nonzero((Xk == pb[s, :3]).all(axis=1))[0]
        if len(ia) == 0 or len(ib) == 0 or abs(int(ia[0]) - int(ib[0])) != 1:
            return False
    return True


def project_xy(M, pts):
    """ndc xy of pts [P,3] under a row-vector 4x4 projection (pytorch3d Transform3d.transform_points: [x y z 1] M, / w),
    fp32 with the evaluation order of the HIP kernels."""
    F = np.float32
    pts = np.asarray(pts, F)
    M = np.asarray(M, F).reshape(4, 4)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    lin = lambda j: ((x * M[0, j] + y * M[1, j]) + z * M[2, j]) + M[3, j]
    with np.errstate(divide="ignore", invalid="ignore"):
        return (lin(0) / lin(3)).astype(F), (lin(1) / lin(3)).astype(F)


def filter_proxy_points(proj, X, pc, filter_tol=0.01):
    """Restates macarons/utility/scone_utils.py:1001-1027 given the views' full-projection matrices proj [n_view,4,4]:
    mask = AND over views and over {x, y} of (X_proj < max(pc_proj) + tol) & (X_proj > min(pc_proj) - tol).  Returns
    (mask bool [P], bounds [n_view,4] = min_x, max_x, min_y, max_y)."""
    F = np.float32
    proj = np.asarray(proj, F)
    mask = np.ones(len(X), bool)
    bounds = np.zeros((proj.shape[0], 4), F)
    tol = F(filter_tol)
    for v in range(proj.shape[0]):
        cx, cy = project_xy(proj[v], pc)
        nx, ny = project_xy(proj[v], X)
        bounds[v] = [cx.min(), cx.max(), cy.min(), cy.max()]
        mask &= (nx < F(bounds[v, 1] + tol)) & (nx > F(bounds[v, 0] - tol)) & (ny < F(bounds[v, 3] + tol)) & (ny > F(bounds[v, 2] - tol))
    return mask, bounds


def view_space_bin_indices(X_cam_inv, n_elev, n_azim):
    """scone_utils.py:901-926 in fp32: bin index of each grid direction once moved to the camera's view space."""
    X = np.asarray(X_cam_inv, F).reshape(-1, 3)
    elev_step, azim_step = F(np.pi / (n_elev + 1)), F(2 * np.pi / n_azim)
    _, ray_elev, ray_azim = sh.spherical_coords(X)
    idx_elev = floor_divide(ray_elev, elev_step)
    idx_azim = floor_divide(ray_azim, azim_step)
    idx_elev[np.mod(ray_elev, elev_step) > F(np.pi / (n_elev + 1) / 2.)] += 1
    idx_azim[np.mod(ray_azim, azim_step) > F(2 * np.pi / n_azim / 2.)] += 1
    idx_elev[idx_elev > n_elev // 2] = n_elev // 2
    idx_elev[idx_elev < -(n_elev // 2)] = -(n_elev // 2)         # parenthesised here, unlike compute_view_state (:916-917 vs :839)
    idx_azim[idx_azim > n_azim // 2] = -(n_azim // 2)
    idx_elev += n_elev // 2
    idx_azim[idx_azim < 0] += n_azim
    return idx_elev.astype(np.int64) * n_azim + idx_azim.astype(np.int64)


def view_space_grid(n_elev, n_azim):
    """The unit grid directions X_cam_ref of scone_utils.py:880-895 (elevation-major)."""
    elev = np.array([-90. + (i + 1) / (n_elev + 1) * 180. for i in range(n_elev) for j in range(n_azim)], F)
    azim = np.array([360. * j / n_azim for i in range(n_elev) for j in range(n_azim)], F)
    return sh.cartesian_coords(np.ones(len(elev), F), elev, azim, in_degrees=True)


def move_view_state_to_view_space(view_state, X_cam_inv, n_elev, n_azim):
    """scone_utils.py:863-931 given the grid directions already moved to view space (the camera transform is PyTorch3D's)."""
    idx = view_space_bin_indices(X_cam_inv, n_elev, n_azim)
    return np.asarray(view_state)[..., idx], idx


def view_space_bin_margin(X_cam_inv, n_elev, n_azim):
    """Angular distance (rad) of each moved grid direction from the nearest decision boundary of the binning above (the
    half-way points between bins and the +-180 degree wrap): closer than a few ulps of asin/acos, the bin depends on the libm
    at hand (an axis-aligned camera puts a dozen of the 98 directions exactly there)."""
    X = np.asarray(X_cam_inv, np.float64).reshape(-1, 3)
    r = np.linalg.norm(X, axis=1)
    elev = np.arcsin(np.clip(X[:, 1] / r, -1, 1))
    azim = np.arctan2(X[:, 0], X[:, 2])
    es, az = np.pi / (n_elev + 1), 2 * np.pi / n_azim
    fe = np.abs(((elev / es) - 0.5) - np.round((elev / es) - 0.5)) * es
    fa = np.abs(((azim / az) - 0.5) - np.round((azim / az) - 0.5)) * az
    pole = np.hypot(X[:, 0], X[:, 2]) / r                                # azimuth is ill-defined at the poles
    return np.minimum(np.minimum(fe, fa), np.where(pole < 1e-6, 0.0, np.inf))
