"""One SCONE NBV decision composed from the oracle pieces (numpy).  TEST INFRASTRUCTURE ONLY.
Restates the step of macarons/testers/shapenet.py:126-172 (without the PyTorch3D camera objects)."""
import numpy as np

from . import nets, scorer, view_state as V


def nbv_step(sd_occ, sd_vis, pc, X, X_view, X_cam, occ_perms, samples, n_elev=7, n_azim=14, min_occ=0.1, dtype=np.float32):
    base, h_polar, h_azim = V.all_harmonics_under_degree(8, n_elev, n_azim)
    vs = V.compute_view_state(X, X_view, n_elev, n_azim)
    vh = V.compute_view_harmonics(vs, base, h_polar, h_azim, n_elev, n_azim)
    occ = nets.scone_occ_forward(sd_occ, pc, X, vh, occ_perms, dtype).reshape(-1, 1).astype(np.float32)
    res, res_h, inv, _ = V.sample_proxy_points(X[0], occ, vh[0], samples, min_occ, exact=True)
    harm = nets.scone_vis_forward(sd_vis, res[None], res_h[None], dtype)
    pts = res[inv][None]
    harm = harm[0][inv][None]
    gains = scorer.compute_coverage_gain(pts, harm, X_cam[None], True, "trigfree", np.float64)
    return {"occ": occ, "gains": gains[0], "nbv_idx": int(np.argmax(gains[0])), "n_unique": len(res), "view_harmonics": vh}
