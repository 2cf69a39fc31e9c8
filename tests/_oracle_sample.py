"""Bounded oracle checks for NBV decisions at sizes the oracle cannot run whole (BASELINE headline Q = 100k; config 3: 8 x 32k).

SconeOcc queries are independent given the surface cloud and the hidden draws, the view harmonics of a row depend on its point
only, and the scorer is per camera: so a SAMPLE of the queries, the (<= 2048-point) sampled set and a handful of cameras can be
recomputed by the numpy / C oracle in seconds and compared with what the full-size HIP step returned.  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import torch

from oracle import cport, nets
from oracle import view_state as V


def check_step_against_oracle(r, sdo, sdv, pc, X, X_view, cams, perms, u, n_q=64, n_cam=7, min_occ=0.1, seed=0, n_elev=7, n_azim=14):
    """r: the dict nbv_step(..., return_samples=True) returned for ONE cloud (occ [Q,1], gains [C], proxy_points [seq_len,4],
    sample_idx [seq_len], n_unique); pc [1,M,3], X [1,Q,3], X_view [n_view,3], cams [C,3], u [seq_len] numpy arrays; perms: the three
    index arrays.  Asserts, all at 1e-4 relative (indices / point sets exact):
      1. the view harmonics the HIP kernels produce for a sample of the queries (first, last, n_q random) == oracle/view_state.py;
      2. the occupancies of those queries == oracle.nets.scone_occ_forward on the sample (same cloud, same draws);
      3. the sampled set: the oracle sampler (fp64 CDF) run on the HIP occupancies of ALL queries picks the same points / inverse map;
      4. the gains of n_cam cameras == oracle SconeVis on the sampled set -> the C port of the reference scorer."""
    from macarons_amd.utility import scone_utils as su
    dev = r["occ"].device
    Q = X.shape[1]
    rng = np.random.default_rng(seed)
    idx = np.unique(np.concatenate(([0, Q - 1], rng.choice(Q, n_q, replace=False))))
    Xs = np.ascontiguousarray(X[:, idx])
    base, h_polar, h_azim = V.all_harmonics_under_degree(8, n_elev, n_azim)
    b_h, p_h, a_h = su.get_all_harmonics_under_degree(8, n_elev, n_azim, dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def harmonics_of(P3, what):
        """Oracle view harmonics of the points P3 [1,n,3], checked against the HIP kernels' on every row whose rays are clear of the
        bin boundaries (a ray within fp32 rounding of a boundary is legitimately ambiguous between asin / acos implementations: those
        rows -- a handful at most -- take the HIP value, so that what follows checks the networks and not the ambiguity)."""
        clear = (V.bin_boundary_margin(P3, X_view, n_elev, n_azim) > 1e-5).all(-1)[0]
        vh_or = V.compute_view_harmonics(V.compute_view_state(P3, X_view, n_elev, n_azim), base, h_polar, h_azim, n_elev, n_azim)
        vh_hip = su.compute_view_harmonics(su.compute_view_state(T(P3), T(X_view), n_elev, n_azim), b_h, p_h, a_h, n_elev, n_azim).cpu().numpy()
        assert clear.sum() >= P3.shape[1] - 4, what
        assert np.abs(vh_hip[0][clear] - vh_or[0][clear]).max() < 1e-5 * max(np.abs(vh_or).max(), 1e-30), what
        return np.where(clear[None, :, None], vh_or, vh_hip)
    # 1 ---- view harmonics of the sampled rows; 2 ---- their occupancies
    vh_in = harmonics_of(Xs, "view harmonics of the sampled queries")
    occ_or = nets.scone_occ_forward(sdo, pc, Xs, vh_in, [np.asarray(p) for p in perms]).reshape(-1)
    occ_hip = r["occ"].reshape(-1).cpu().numpy()
    scale = np.abs(occ_or).max()
    assert np.abs(occ_hip[idx] - occ_or).max() < 1e-4 * scale, ("occupancy sample", np.abs(occ_hip[idx] - occ_or).max(), scale)
    # 3 ---- the sampled set from the HIP occupancies
    nu = int(r["n_unique"])
    pp = r["proxy_points"].cpu().numpy()
    inv = r["sample_idx"].cpu().numpy()
    res, _, inv_or, uniq_or = V.sample_proxy_points(X[0], occ_hip[:, None], np.zeros((Q, 1), np.float32), u, min_occ, exact=True)
    assert nu == len(res) and np.array_equal(pp[:nu], res) and not pp[nu:].any() and np.array_equal(inv, inv_or), "sampled set"
    # 4 ---- SconeVis on the sampled set (oracle) -> C port of the reference scorer, n_cam cameras
    vh_s = harmonics_of(np.ascontiguousarray(res[None, :, :3]), "view harmonics of the sampled set")
    harm = nets.scone_vis_forward(sdv, res[None], vh_s)
    C = cams.shape[0]
    cam_idx = np.unique(np.concatenate(([0, C - 1, int(r["nbv_idx"])], rng.choice(C, max(n_cam - 3, 1), replace=False))))
    g_or, _ = cport.coverage_gain(res[inv_or][None], harm[0][inv_or][None], np.ascontiguousarray(cams[cam_idx])[None])
    g_hip = r["gains"].cpu().numpy()[cam_idx]
    assert np.abs(g_hip - g_or[0]).max() < 1e-4 * np.abs(g_or).max(), ("gains", g_hip, g_or[0])
    return {"queries": len(idx), "cams": len(cam_idx), "occ_err": float(np.abs(occ_hip[idx] - occ_or).max() / scale),
            "gain_err": float(np.abs(g_hip - g_or[0]).max() / np.abs(g_or).max())}
