"""`torch.ops.macarons.*` — the operator names SURVEY §8(b) lists for the layer below the Python class boundary, registered
with `torch.library` on the CUDA (= HIP) dispatch key.  Every operator is a thin forwarder to `macarons_amd.ops`, i.e. to the
C ABI of `include/macarons_hip.h`; there is no CPU implementation (calling one with CPU tensors raises torch's own
"no kernel for backend CPU" error — the MI355X path has no fallback).  Importing this module performs the registration:

    import macarons_amd.torch_ops                      # once
    gains = torch.ops.macarons.sh_coverage_gain(pts, harmonics, cams, True)

Tensor lists (`weights`, `pc_scales`, `local_blobs`) are what `SconeVis.weight_table()` / `SconeOcc.weight_table()` /
`networks.packing.pack_local_pct` return.  All operators launch on the current HIP stream and never synchronise, except
`sample_proxy`, whose unique count sizes its outputs (the reference's `torch.unique` synchronises at the same point).
"""
import torch

from . import ops

_NS = "macarons"
_lib = torch.library.Library(_NS, "DEF")

_SCHEMAS = {
    "sh_coverage_gain": "(Tensor pts, Tensor harmonics, Tensor cams, bool use_sigmoid) -> Tensor",
    "sh_visibilities": "(Tensor pts, Tensor harmonics, Tensor cams, bool use_sigmoid) -> Tensor",
    "knn_gather_offset": "(Tensor x, Tensor pc, int k) -> (Tensor, Tensor, Tensor)",
    "points_in_fov": "(Tensor pts, Tensor cameras) -> Tensor",
    "view_state": "(Tensor pts, Tensor X_view, int n_elev, int n_azim) -> Tensor",
    "view_harmonics": "(Tensor view_state, Tensor matrix) -> Tensor",
    "sample_proxy": "(Tensor X, Tensor probs, Tensor view_harmonics, Tensor u, float min_occ) -> (Tensor, Tensor, Tensor, Tensor)",
    "scone_vis_forward": "(Tensor pts, Tensor view_harmonics, Tensor[] weights) -> Tensor",
    "scone_occ_forward": "(Tensor pc_global, Tensor[] pc_scales, Tensor x, Tensor view_harmonics, Tensor[] weights, Tensor[] local_blobs) -> Tensor",
}
for _name, _schema in _SCHEMAS.items():
    _lib.define(_name + _schema)


def _impl(name):
    def deco(fn):
        _lib.impl(name, fn, "CUDA")
        return fn
    return deco


@_impl("sh_coverage_gain")
def _sh_coverage_gain(pts, harmonics, cams, use_sigmoid):
    """SconeVis.compute_coverage_gain (SconeVis.py:210-252): pts [B,N,3|4], harmonics [B,N,64], cams [B,C,3] -> [B,C]."""
    return ops.sh_coverage_gain(pts, harmonics, cams, use_sigmoid)


@_impl("sh_visibilities")
def _sh_visibilities(pts, harmonics, cams, use_sigmoid):
    """SconeVis.compute_visibilities (SconeVis.py:164-208) -> [B,C,N]."""
    return ops.sh_visibilities(pts, harmonics, cams, use_sigmoid)


@_impl("knn_gather_offset")
def _knn_gather_offset(x, pc, k):
    """get_knn_points + the offset step (utils.py:1497-1509, SconeOcc.py:297-298) -> (offsets [B,Q,k,3], dists, idx int64)."""
    return ops.knn_points(x, pc, k, True)


@_impl("points_in_fov")
def _points_in_fov(pts, cameras):
    """Camera.get_points_in_fov (macarons_utils.py:2400-2435): pts [P,3], 40-float camera records [n_cam,40] -> bool [n_cam,P]."""
    return ops.points_in_fov(pts, cameras)


@_impl("view_state")
def _view_state(pts, X_view, n_elev, n_azim):
    """compute_view_state (scone_utils.py:799-860) -> [B,Q,n_elev*n_azim]."""
    return ops.view_state(pts, X_view, n_elev, n_azim)


@_impl("view_harmonics")
def _view_harmonics(view_state, matrix):
    """compute_view_harmonics (scone_utils.py:934-960) as one product with the constant [n_harmonics, n_bins] matrix."""
    return ops.linear(view_state, matrix)


@_impl("sample_proxy")
def _sample_proxy(X, probs, view_harmonics, u, min_occ):
    """sample_proxy_points (scone_utils.py:1030-1061) -> (points+occupancy [n_u,4], harmonics [n_u,64], inverse, unique idx)."""
    return tuple(ops.sample_proxy(X, probs, view_harmonics, u, min_occ))


@_impl("scone_vis_forward")
def _scone_vis_forward(pts, view_harmonics, weights):
    """SconeVis.forward (SconeVis.py:121-162)."""
    return ops.scone_vis_forward(pts, view_harmonics, list(weights))


@_impl("scone_occ_forward")
def _scone_occ_forward(pc_global, pc_scales, x, view_harmonics, weights, local_blobs):
    """SconeOcc.forward (SconeOcc.py:250-347) after the three down-sampling draws (pc_global, the three scale clouds)."""
    return ops.scone_occ_forward(pc_global, list(pc_scales), x, view_harmonics, list(weights), list(local_blobs) or None)


def registered():
    """Names of the operators under torch.ops.macarons."""
    return sorted(_SCHEMAS)
