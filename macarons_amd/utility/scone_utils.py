"""Mirror of the hot-path helpers of macarons/utility/scone_utils.py — same names, arguments and tensor layouts —
backed by the MI355X kernels (include/macarons_hip.h).  Reference lines:
  get_all_harmonics_under_degree :714   get_cameras_on_sphere :741   normalize_points_in_prediction_box :788
  compute_view_state :799   compute_view_harmonics :934   compute_occupancy_probability :965
  sample_proxy_points :1030
(filter_proxy_points :1001 needs PyTorch3D camera objects; filter_proxy_points_from_matrices takes the
projection matrices instead — camera objects stay outside the kernels)
"""
import numpy as np
import torch

from .. import ops
from .CustomGeometry import get_cartesian_coords, get_spherical_coords
from .spherical_harmonics import get_spherical_harmonics


def _lattice(n_elev, n_azim, elev_span, azim_span):
    """The (elevation-major) direction lattice both helpers below share: row i sits at -span/2 + (i + 1) / (n_elev + 1) * span,
    column j at azim_span * j / n_azim.  Evaluated in float64 in the reference's operation order and rounded to float32 once,
    so the values are the reference's bit for bit (scone_utils.py:724-727, 765-771)."""
    rows = (torch.arange(n_elev, dtype=torch.float64) + 1.0) / (n_elev + 1) * elev_span - elev_span / 2
    cols = azim_span * torch.arange(n_azim, dtype=torch.float64) / n_azim
    return rows.repeat_interleave(n_azim).float(), cols.repeat(n_elev).float()


_harm_tables = {}


def get_all_harmonics_under_degree(degree, n_elev, n_azim, device):
    """-> (z [degree^2, n_elev*n_azim], h_polar, h_azim); elevation-major grid (scone_utils.py:714-738).  The lattice is a constant
    of (degree, n_elev, n_azim): built once per device (upstream rebuilds it -- ~70 small launches -- on every call)."""
    key = (degree, n_elev, n_azim, str(device))
    hit = _harm_tables.get(key)
    if hit is None:
        hit = _harm_tables[key] = _all_harmonics_under_degree(degree, n_elev, n_azim, device)
    return hit


def _all_harmonics_under_degree(degree, n_elev, n_azim, device):
    h_elev, h_azim = (t.to(device) for t in _lattice(n_elev, n_azim, np.pi, 2 * np.pi))
    h_polar = -h_elev + np.pi / 2
    z = torch.cat([get_spherical_harmonics(l, h_polar, h_azim) for l in range(degree)], dim=-1)
    return z.transpose(dim0=0, dim1=1).contiguous(), h_polar, h_azim


def get_cameras_on_sphere(params=None, device="cpu", pole_cameras=False, n_elev=None, n_azim=None, camera_dist=None):
    """(X_cam [n,3], dist, elev, azim): candidate cameras on the reference's sphere lattice, optionally with the two near-pole
    cameras at +-89.9 degrees in front and behind (scone_utils.py:741-785)."""
    if n_elev is None or n_azim is None:
        n_elev, n_azim = params.n_camera_elev, params.n_camera_azim
    if camera_dist is None:
        camera_dist = params.camera_dist
    elev, azim = _lattice(n_elev, n_azim, 180.0, 360.0)
    if pole_cameras:
        elev = torch.cat((elev.new_tensor([-89.9]), elev, elev.new_tensor([89.9])))
        azim = torch.cat((azim.new_zeros(1), azim, azim.new_zeros(1)))
    elev, azim = elev.to(device), azim.to(device)
    dist = torch.full_like(elev, float(camera_dist))
    X_cam = get_cartesian_coords(r=dist.view(-1, 1), elev=elev.view(-1, 1), azim=azim.view(-1, 1), in_degrees=True)
    return X_cam, dist, elev, azim


def normalize_points_in_prediction_box(points, prediction_box_center, prediction_box_diag):
    return (points - prediction_box_center) / prediction_box_diag


def compute_view_state(pts, X_view, n_elev, n_azim):
    """pts [n_cloud, seq_len, >=3], X_view [n_view, 3] -> [n_cloud, seq_len, n_elev*n_azim]  (scone_utils.py:799-860)."""
    return ops.view_state(pts, X_view, n_elev, n_azim)


_vh_cache = {}


def _view_harmonics_matrix(base_harmonics, h_polar, n_elev, n_azim):
    key = (base_harmonics.data_ptr(), base_harmonics._version, h_polar.data_ptr(), n_elev, n_azim)
    m = _vh_cache.get(key)
    if m is None:
        polar_step, azim_step = np.pi / (n_elev + 1), 2 * np.pi / n_azim
        # the constant factor of scone_utils.py:958, in the reference's multiplication order
        m = (base_harmonics * torch.sin(h_polar).view(1, -1) * polar_step * azim_step).float().contiguous()
        _vh_cache.clear()
        _vh_cache[key] = m
    return m


def compute_view_harmonics(view_state, base_harmonics, h_polar, h_azim, n_elev, n_azim):
    """[n_cloud, seq_len, n_elev*n_azim] -> [n_cloud, seq_len, n_harmonics]  (scone_utils.py:934-960): one
    [.,98] x [98,64] product instead of the reference's [., 64, 98] broadcast."""
    return ops.linear(view_state, _view_harmonics_matrix(base_harmonics, h_polar, n_elev, n_azim))


def compute_occupancy_probability(scone_occ, pc, X, view_harmonics, mask=None, max_points_per_pass=20000):
    """Chunked occupancy inference with the reference's chunk boundaries (scone_utils.py:965-998), so the hidden
    randperm draws of SconeOcc.forward happen once per chunk exactly like upstream."""
    n_clouds = pc.shape[0]
    n_sample = X.shape[1]
    p = max_points_per_pass // n_clouds
    q, r = n_sample // p, n_sample % p
    n_loop = q + (1 if r != 0 else 0)
    if n_loop == 0:                                     # no query point: the reference returns an empty [n_clouds, 0, 1] tensor (:980)
        return X.new_zeros(n_clouds, 0, 1)
    preds = []
    for i in range(n_loop):
        low, up = i * p, (i + 1) * p
        if i == q:
            up = q * p + r
        preds.append(scone_occ(pc, X[:, low:up].contiguous(), view_harmonics[:, low:up].contiguous(), verbose=False)
                     .view(n_clouds, up - low, -1))
    return preds[0] if len(preds) == 1 else torch.cat(preds, dim=1)


def view_space_bin_indices(X_cam_inv, n_elev, n_azim):
    """Bin index of each rotated grid direction (scone_utils.py:901-926): nearest (elevation, azimuth) bin, elevation clamped to
    +-(n_elev//2), azimuth wrapped at 180 degrees.  X_cam_inv [n_elev*n_azim, 3] fp32 (any device) -> int64 indices (CPU)."""
    X = X_cam_inv.detach().to("cpu", torch.float32).view(-1, 3)
    elev_step, azim_step = np.pi / (n_elev + 1), 2 * np.pi / n_azim
    _, ray_elev, ray_azim = get_spherical_coords(X)
    fd = lambda a, st: (a - a % st) / st                                # utils.floor_divide (utils.py:113-117)
    idx_elev, idx_azim = fd(ray_elev, elev_step), fd(ray_azim, azim_step)
    idx_elev = idx_elev + (ray_elev % elev_step > elev_step / 2.).to(idx_elev.dtype)
    idx_azim = idx_azim + (ray_azim % azim_step > azim_step / 2.).to(idx_azim.dtype)
    idx_elev = idx_elev.clamp(-(n_elev // 2), n_elev // 2)
    idx_azim = torch.where(idx_azim > n_azim // 2, torch.full_like(idx_azim, -(n_azim // 2)), idx_azim)
    idx_elev = idx_elev + n_elev // 2
    idx_azim = torch.where(idx_azim < 0, idx_azim + n_azim, idx_azim)
    return idx_elev.long() * n_azim + idx_azim.long()


_REF_DIRECTIONS = {}
_NATIVE_BINS = []


def _native_bins():
    """Is torch.ops.macarons.view_space_bins there (the C++ extension built)?  MCR_NATIVE_BINS=0: the Python restatement (A/B, tests)."""
    if not _NATIVE_BINS:
        import os
        ok = os.environ.get("MCR_NATIVE_BINS", "1") != "0"
        if ok:
            try:
                from .. import torch_ops  # noqa: F401
                ok = hasattr(torch.ops.macarons, "view_space_bins")
            except Exception:
                ok = False
        _NATIVE_BINS.append(ok)
    return _NATIVE_BINS[0]


def view_space_bin_permutation(fov_camera, n_elev, n_azim, device="cpu"):
    """The bin permutation of move_view_state_to_view_space (scone_utils.py:863-931) as int64 indices on the host: column v of the
    rotated view state comes from bin indices[v].  `fov_camera`: see move_view_state_to_view_space; `device`: where a camera object's
    transform runs."""
    X_ref = _REF_DIRECTIONS.get((n_elev, n_azim))
    if X_ref is None:                                   # the lattice's unit directions: a constant of (n_elev, n_azim)
        n_view = n_elev * n_azim
        elev = torch.Tensor([-90. + (i + 1) / (n_elev + 1) * 180. for i in range(n_elev) for j in range(n_azim)])
        azim = torch.Tensor([360. * j / n_azim for i in range(n_elev) for j in range(n_azim)])
        X_ref = _REF_DIRECTIONS[(n_elev, n_azim)] = get_cartesian_coords(r=torch.ones(n_view, 1), elev=elev.view(-1, 1), azim=azim.view(-1, 1),
                                                                           in_degrees=True)
    if torch.is_tensor(fov_camera):
        R = fov_camera.detach().to("cpu", torch.float32)
        if _native_bins():                              # the same ATen operators from C++ (one dispatcher call instead of ~40: this host work
            return torch.ops.macarons.view_space_bins(X_ref, R, n_elev, n_azim)     # sits on a MACARONS decision's critical path)
        X_inv = X_ref @ R.view(3, 3).T
    else:
        X_inv = fov_camera.get_world_to_view_transform().inverse().transform_points(X_ref.to(device)) - fov_camera.get_camera_center()
    return view_space_bin_indices(X_inv.reshape(-1, 3), n_elev, n_azim)


def move_view_state_to_view_space(view_state, fov_camera, n_elev, n_azim):
    """"Rotate" the view-state vectors into a camera's view space (scone_utils.py:863-931): view_state [n_cloud, seq_len,
    n_elev*n_azim] -> same shape, column v taken from the bin the v-th grid direction lands in after the inverse
    world-to-view transform.  `fov_camera`: the reference's camera object (its get_world_to_view_transform().inverse()
    .transform_points and get_camera_center are called exactly as the reference does; PyTorch3D stays outside the kernels)
    or the 3x3 world-to-view rotation R of the row-vector convention X_view = X_world R + T (a CPU tensor costs nothing; a device
    tensor is read back)."""
    indices = view_space_bin_permutation(fov_camera, n_elev, n_azim, view_state.device)
    return ops.gather_columns(view_state.contiguous(), indices)


def filter_proxy_points(view_cameras, X, pc, filter_tol=0.01):
    """(X[mask], mask): scone_utils.py:1001-1027.  `view_cameras` is either the reference's PyTorch3D camera batch (anything
    with get_full_projection_transform().get_matrix() -> [n_view,4,4], row-vector convention) or that matrix stack itself;
    PyTorch3D objects stay outside the kernels.  Works for a single scene: X [P,3], pc [N,3]."""
    if (len(X.shape) != 2) or (len(pc.shape) != 2):
        raise NameError("Wrong shapes! X must have shape (n_proxy_points, 3) and pc must have shape (N, 3).")
    proj = view_cameras if torch.is_tensor(view_cameras) else view_cameras.get_full_projection_transform().get_matrix()
    proj = proj.to(device=X.device, dtype=torch.float32).contiguous()
    mask, _ = ops.filter_proxy_mask(X.contiguous(), pc.contiguous(), proj, filter_tol)
    return X[mask], mask


def sample_proxy_points(X_world, preds, view_harmonics, n_sample, min_occ, use_occ_to_sample=True, return_index=False,
                        samples=None, padded=False):
    """scone_utils.py:1030-1076.  `samples` (optional, [n_sample]) pins the uniforms; otherwise they are drawn with
    torch.rand(n_sample, 1, device=...) like the reference (:1052).  padded=True (extension): no host sync -- returns
    (res [n_sample,4], res_h [n_sample,64], inverse_idx, n_unique int32 device tensor [1]) with zero rows beyond n_unique."""
    if not use_occ_to_sample:
        mask = preds[..., 0] > min_occ
        res_X, res_preds, res_h = X_world[mask][:n_sample], preds[mask][:n_sample], view_harmonics[mask][:n_sample]
        res = torch.cat((res_X, res_preds), dim=-1)
        return (res, res_h, None) if return_index else (res, res_h)
    if samples is None:
        samples = torch.rand(n_sample, 1, device=X_world.device)
    if padded:
        res, res_h, inverse_idx, _, nu = ops.sample_proxy(X_world, preds.reshape(-1), view_harmonics, samples.reshape(-1), min_occ,
                                                          padded=True)
        return res, res_h, inverse_idx, nu
    res, res_h, inverse_idx, _ = ops.sample_proxy(X_world, preds.reshape(-1), view_harmonics, samples.reshape(-1), min_occ)
    return (res, res_h, inverse_idx) if return_index else (res, res_h)
