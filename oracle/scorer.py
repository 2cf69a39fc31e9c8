"""Per-candidate-camera surface-coverage-gain scorer (numpy restatement).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates:
  * macarons/networks/SconeVis.py:164-208  compute_visibilities   -> [B,C,N]
  * macarons/networks/SconeVis.py:210-252  compute_coverage_gain  -> [B,C]
  * macarons/networks/SconeVis.py:254-303  compute_coverage_gain_multiple
  * macarons/networks/Macarons.py:138-178  compute_visibility_gains (== compute_visibilities)
"""
import itertools
import numpy as np
from . import sh


def _z(pts, harmonics, X_cam, style, dtype, cam_chunk=16):
    pts = np.asarray(pts)
    harmonics = np.asarray(harmonics)
    X_cam = np.asarray(X_cam)
    B, N = pts.shape[0], pts.shape[1]
    C = X_cam.shape[1]
    X_pts = pts[..., :3].astype(dtype)                        # SconeVis.py:224
    h = harmonics.astype(dtype)
    out = np.empty((B, C, N), dtype=dtype)
    for b in range(B):
        for c0 in range(0, C, cam_chunk):
            c1 = min(C, c0 + cam_chunk)
            rays = (X_cam[b, c0:c1, None, :].astype(dtype) - X_pts[b][None, :, :]).reshape(-1, 3)   # :230-231
            if style == "literal":
                _, elev, phi = sh.spherical_coords(rays)                                               # :232
                theta = (-elev + dtype(np.pi / 2.0)).astype(dtype)                                     # :233
                Y = sh.sh_basis_literal(theta, phi, dtype)                                             # :235-239
            else:
                Y = sh.sh_basis_trigfree(rays, dtype)
            Y = Y.reshape(c1 - c0, N, sh.N_HARMONICS)
            out[b, c0:c1] = (Y * h[b][None]).sum(axis=-1)                                             # :241
    return out


def compute_visibilities(pts, harmonics, X_cam, use_sigmoid=True, style="literal", dtype=np.float32):
    z = _z(pts, harmonics, X_cam, style, dtype)
    if use_sigmoid:
        with np.errstate(over="ignore"):
            z = (1.0 / (1.0 + np.exp(-z))).astype(dtype)      # :242-243
    else:
        z = np.maximum(z, 0).astype(dtype)                     # :244-245
    return z


def compute_coverage_gain(pts, harmonics, X_cam, use_sigmoid=True, style="literal", dtype=np.float32):
    z = compute_visibilities(pts, harmonics, X_cam, use_sigmoid, style, dtype)
    return (z.sum(axis=-1, dtype=dtype) / dtype(z.shape[-1])).astype(dtype)   # :250


def compute_coverage_gain_multiple(pts, harmonics, X_cam, n_cam, use_sigmoid=True, style="literal",
                                   dtype=np.float32):
    """SconeVis.py:254-303: every ordered n_cam-tuple of cameras; mean over points of max over the tuple."""
    if n_cam not in (2, 3):
        raise NameError("n_cam is too large.")                  # :298
    z = compute_visibilities(pts, harmonics, X_cam, use_sigmoid, style, dtype)   # [B,C,N]
    C = z.shape[1]
    n_idx = np.array(list(itertools.product(range(C), repeat=n_cam)), dtype=np.int64)   # cartesian_prod :294-296
    n_z = z[:, n_idx]                                            # [B, C^n, n, N]
    n_z = n_z.max(axis=-2).sum(axis=-1, dtype=dtype) / dtype(z.shape[-1])
    return n_z.astype(dtype), n_idx
