"""HIP path vs goldens the REFERENCE's own functions produced for the MACARONS-regime rows:
Camera.get_points_in_fov on a real Camera object, the three distance factors, predict_coverage_gain_for_single_camera,
Cell.fill, depth unprojection and the Macarons wrapper (tests/golden/make_golden.py: gen_fov, gen_distance,
gen_single_camera, gen_cell, gen_unproject, gen_macarons_wrapper).  Masks and kept point sets bit-exact, gains at 1e-4."""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from conftest import golden, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import weights  # noqa: E402

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _records(g, fov_range):
    from macarons_amd.utility.macarons_utils import camera_record
    return torch.stack([camera_record(g["Mview"][c], g["Mfull"][c], g["ndc"], g["center"][c], fov_range) for c in range(len(g["eyes"]))])


def _models(dev):
    from macarons_amd.networks import SconeVis, SconeOcc, Macarons
    occ, vis = SconeOcc(), SconeVis()
    sdo = weights.make_state_dict(weights.shapes_of(occ), 2)
    sdv = weights.make_state_dict(weights.shapes_of(vis), 1)
    sdo["linear3.bias"] = sdo["linear3.bias"] + np.float32(0.5)
    occ.load_state_dict({k: torch.from_numpy(v) for k, v in sdo.items()})
    vis.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()})
    return Macarons(None, occ, vis).to(dev).eval()


def test_points_in_fov_matches_reference_camera(dev):
    from macarons_amd import ops
    g = golden("fov_camera")
    n = len(g["pts"])
    for tag, rg in (("r40", 40.0), ("none", None)):
        mask = ops.points_in_fov(T(g["pts"], dev), _records(g, rg).to(dev)).cpu().numpy()
        for c in range(len(g["eyes"])):
            ref = np.unpackbits(g[f"mask_{c}_{tag}"])[:n].astype(bool)
            assert np.array_equal(mask[c], ref), (tag, c)


def test_distance_factors_match_reference(dev):
    from macarons_amd.utility import macarons_utils as mu
    g = golden("distance_factors")
    params = NS(image_height=int(g["hw"][0]), image_width=int(g["hw"][1]))
    fc = NS(fov=torch.tensor([float(g["fov"])]))
    pts, cam, res = T(g["pts"], dev), T(g["cam"], dev), float(g["cell_resolution"])
    assert rel_err(mu.get_distance_factor_threshold(pts, cam, 17.).cpu().numpy(), g["f_th"]) < 1e-6
    assert rel_err(mu.get_distance_factor(params, pts, cam, fc, res).cpu().numpy(), g["f_plain"]) < 2e-6
    assert rel_err(mu.get_distance_factor_smooth(params, pts, cam, fc, res).cpu().numpy(), g["f_smooth"]) < 2e-6


def test_single_camera_gain_matches_reference(dev):
    """predict_coverage_gain_for_cameras (all neighbour cameras in one call) vs predict_coverage_gain_for_single_camera of the
    reference, camera by camera: gains 1e-4, identical sampled world points, the empty frustum scores 0."""
    from macarons_amd.utility import macarons_utils as mu
    g = golden("single_camera")
    m = _models(dev)
    P_ = len(g["X_world"])
    vh = (g["vh_u"][:, None] * g["vh_v"][None, :] + g["vh_w"][np.arange(P_) % 16]).astype(np.float32)
    recs = _records(g, float(g["sensor_range"])).to(dev)
    u = torch.zeros(4, 2048, device=dev)
    for c in range(3):
        u[c] = T(g[f"u_{c}"], dev)
    Mpred = T(np.repeat(g["Mpred"], 4, 0), dev)
    args = (m.visibility, T(g["X_world"], dev), T(vh, dev), T(g["occ"], dev), recs, T(g["eyes"], dev), Mpred, float(g["box_diag"]))
    gains, vis, world = mu.predict_coverage_gain_for_cameras(*args, samples=u, return_parts=True)
    gains = gains.cpu().numpy()
    for c in range(3):
        ref = float(g[f"gain_{c}"].ravel()[0])
        assert abs(gains[c] - ref) < 1e-4 * ref, (c, gains[c], ref)
        assert np.array_equal(world[c].cpu().numpy(), g[f"world_{c}"][0])
        # per-point gains vs the reference's modules run in float64 on the same sampled set (its fp32 run is itself off by up to
        # 6e-4: asin -> cos -> acos near the poles)
        assert np.abs(vis[c].cpu().numpy() - g[f"vis64_{c}"][0, 0]).max() < 2e-5
        assert np.abs(vis[c].cpu().numpy() - g[f"vis_{c}"][0, 0]).max() < 2e-3
    assert gains[3] == 0.0 and float(g["gain_3"].ravel()[0]) == 0.0
    # the two other distance-factor branches (:1686-1698)
    params, fc = NS(image_height=256, image_width=456), NS(fov=torch.tensor([60.0]))
    th = mu.sensor_distance_threshold(params, fc, 0.1)
    for tag, smooth in (("plain", False), ("smooth", True)):
        gk = mu.predict_coverage_gain_for_cameras(*[a[:1] if i in (4, 5, 6) else a for i, a in enumerate(args)], samples=u[:1],
                                                  distance_th=th, smooth=smooth).cpu().numpy()
        ref = float(g[f"gain_0_{tag}"].ravel()[0])
        assert abs(gk[0] - ref) < 1e-4 * ref, (tag, gk[0], ref)


def test_cell_fill_matches_reference(dev):
    from macarons_amd.utility import macarons_utils as mu
    g = golden("cell_fill")
    x_min, x_max = T(g["center"] - g["lwh"] / 2, dev), T(g["center"] + g["lwh"] / 2, dev)
    for i in range(3):
        out = mu.cell_fill(T(g[f"before_{i}"], dev), T(g[f"pts_{i}"], dev), x_min, x_max, float(g["resolution"]), int(g["capacity"]),
                           perm=torch.from_numpy(g[f"perm_{i}"].astype(np.int64)))
        assert np.array_equal(out.cpu().numpy(), g[f"after_{i}"]), i


def test_unproject_matches_reference(dev):
    from macarons_amd.utility import macarons_utils as mu
    g = golden("unproject")
    cams = torch.stack([mu.depth_camera_record(g["Mfull"][c], float(g["P"][c, 2, 2]), float(g["P"][c, 3, 2])) for c in range(2)])
    w = mu.project_depth_back_to_3D(T(g["depth"], dev), cams.to(dev)).cpu().numpy()
    assert w.shape == g["world"].shape and rel_err(w, g["world"]) < 1e-5
    H, W = int(g["H"]), int(g["W"])
    mask = torch.from_numpy(np.unpackbits(g["mask"])[:H * W].astype(bool)).to(dev)
    part = mu.compute_partial_point_cloud(T(g["d1"], dev), mask, cams[0], 0.25, fov_range=60.0,
                                          perm=torch.from_numpy(g["perm"].astype(np.int64))).cpu().numpy()
    assert part.shape == g["part"].shape and rel_err(part, g["part"]) < 1e-5


def test_macarons_wrapper_matches_reference(dev):
    """Macarons.forward(mode='occupancy' | 'visibility'), compute_visibility_gains and the NameError paths (Macarons.py:110-178)."""
    g = golden("macarons_wrapper")
    m = _models(dev)
    torch.manual_seed(int(g["seed"]))                       # the hidden randperm draws of SconeOcc.forward come from the CPU generator
    with torch.no_grad():
        o = m(mode='occupancy', partial_point_cloud=T(g["pc"], dev), proxy_points=T(g["x"], dev), view_harmonics=T(g["vh"], dev))
        h = m(mode='visibility', proxy_points=T(g["pts"], dev), view_harmonics=T(g["vh2"], dev))
        v = m.compute_visibility_gains(pts=T(g["pts"], dev), harmonics=T(g["harm"], dev), X_cam=T(g["cams"], dev))
    assert o.shape == g["occ"].shape and rel_err(o.cpu().numpy(), g["occ"]) < 1e-4
    assert rel_err(h.cpu().numpy(), g["harm"]) < 1e-4
    assert v.shape == g["gains64"].shape and np.abs(v.cpu().numpy() - g["gains64"]).max() < 2e-5
    for mode, kw, key in (("occupancy", dict(proxy_points=T(g["x"], dev)), "err_occupancy"),
                          ("visibility", dict(proxy_points=T(g["pts"], dev)), "err_visibility"), ("depth", {}, "err_depth"),
                          ("bogus", {}, "err_bogus")):
        with pytest.raises(NameError) as e:
            m(mode=mode, **kw)
        assert str(e.value) == str(g[key])
    m.visibility.use_sigmoid = False
    with pytest.raises(NameError):
        m.compute_visibility_gains(pts=T(g["pts"], dev), harmonics=T(g["harm"], dev), X_cam=T(g["cams"], dev))


def test_scene_occupancy_field_matches_reference(dev, monkeypatch):
    """compute_scene_occupancy_probability_field on macarons_amd.utility.scene.Scene objects vs the golden the reference function
    produced on its own Scene / Cell objects (tests/golden/make_golden.py: gen_occ_field): identical points in identical order,
    view harmonics 1e-5, occupancies and the updated proxy_proba at 1e-4 (the hidden randperm draws replay from the seed)."""
    from macarons_amd.utility import macarons_utils as mu
    from macarons_amd.utility.scene import Scene
    g = golden("occ_field")
    m = _models(dev)
    n = len(g["proxy"])
    in_fov = torch.from_numpy(np.unpackbits(g["in_fov"])[:n].astype(bool)).to(dev)
    x_min, x_max, grid = T(g["x_min"], dev), T(g["x_max"], dev), [int(v) for v in g["grid"]]
    surface = Scene(x_min, x_max, *grid, cell_capacity=500, cell_resolution=0.2, n_proxy_points=n, device=dev)
    proxy = Scene(x_min, x_max, *grid, cell_capacity=100000, cell_resolution=1e-4, n_proxy_points=n, device=dev, feature_dim=1)
    for i in range(int(g["n_surface_cells"])):
        surface.cells[str([int(v) for v in g[f"cellkey_{i}"]])].cell_pts = T(g[f"cellpts_{i}"], dev)
        c = proxy.cells[str([int(v) for v in g[f"pcellkey_{i}"]])]
        idx = g[f"pcellidx_{i}"].astype(np.int64)
        c.cell_pts, c.cell_features = T(g["proxy"][idx], dev), T(idx.astype(np.float32)[:, None], dev)
    proxy.initialize_proxy_points()
    proxy.proxy_points = T(g["proxy"], dev)
    proxy.proxy_supervision_occ = T(g["sup_occ"].astype(np.float32)[:, None], dev)
    proxy.view_states = T(np.unpackbits(g["view_states"], axis=-1)[:, :98].astype(np.float32), dev)
    proxy.out_of_field = (~in_fov).float().view(-1, 1)
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98)
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        X, H, O = mu.compute_scene_occupancy_probability_field(params, m, None, surface, proxy, dev, prediction_camera=T(g["Mpred"][0], dev))
    assert np.array_equal(X.cpu().numpy(), g["X_world"])
    assert rel_err(H.cpu().numpy(), g["view_harmonics"]) < 1e-5
    scale = np.abs(g["occ_probs"]).max()
    assert np.abs(O.cpu().numpy() - g["occ_probs"]).max() < 1e-4 * scale
    assert np.abs(proxy.proxy_proba.cpu().numpy() - g["proxy_proba"]).max() < 1e-4 * scale
    # the jobs in three groups (the host draws of group g + 1 overlap the GPU work of group g) or in one: the same bits, the same draws
    for n_groups in ("3", "1"):
        monkeypatch.setenv("MCR_FIELD_GROUPS", n_groups)
        torch.manual_seed(int(g["seed"]))
        rec = {}
        with torch.no_grad():
            X2, H2, O2 = mu.compute_scene_occupancy_probability_field(params, m, None, surface, proxy, dev, prediction_camera=T(g["Mpred"][0], dev),
                                                                      record=rec)
        assert torch.equal(X2, X) and torch.equal(H2, H) and torch.equal(O2, O), n_groups
        assert ("groups" in rec["ragged_perms"]) == (n_groups == "3")
        with torch.no_grad():                                   # a repeat with the recorded draws (what the range-guard fallback does)
            _, _, O3 = mu.compute_scene_occupancy_probability_field(params, m, None, surface, proxy, dev, prediction_camera=T(g["Mpred"][0], dev),
                                                                    ragged_perms=rec["ragged_perms"])
        assert torch.equal(O3, O)
    monkeypatch.delenv("MCR_FIELD_GROUPS")
    # the grid bookkeeping itself: cell lookup and Cell.fill through Scene.fill_cells reproduce the reference's cells
    s2 = Scene(x_min, x_max, *grid, cell_capacity=500, cell_resolution=0.2, n_proxy_points=n, device=dev)
    torch.manual_seed(4000)
    s2.fill_cells(T(g["surface"], dev))
    for i in range(int(g["n_surface_cells"])):
        assert np.array_equal(s2.cells[str([int(v) for v in g[f"cellkey_{i}"]])].cell_pts.cpu().numpy(), g[f"cellpts_{i}"]), i


def _decision_scenes(g, dev):
    from macarons_amd.utility.scene import Scene
    n = len(g["proxy"])
    x_min, x_max, grid = T(g["x_min"], dev), T(g["x_max"], dev), [int(v) for v in g["grid"]]
    surface = Scene(x_min, x_max, *grid, cell_capacity=500, cell_resolution=0.2, n_proxy_points=n, device=dev, feature_dim=1)
    torch.manual_seed(5000)
    surface.fill_cells(T(g["surface"], dev), features=torch.zeros(len(g["surface"]), 1, device=dev))
    for i in range(int(g["n_surface_cells"])):                 # Scene.fill_cells reproduces the reference's cells (randperm from the seed)
        assert np.array_equal(surface.cells[str([int(v) for v in g[f"cellkey_{i}"]])].cell_pts.cpu().numpy(), g[f"cellpts_{i}"]), i
    proxy = Scene(x_min, x_max, *grid, cell_capacity=100000, cell_resolution=1e-4, n_proxy_points=n, device=dev, feature_dim=1)
    proxy.initialize_proxy_points()
    proxy.proxy_points = T(g["proxy"], dev)
    return surface, proxy


def test_macarons_decision_matches_reference(dev):
    """Two consecutive NBV decisions of the MACARONS loop (config 5 minus the depth network; testers/scene.py:391-454) through
    macarons_utils.macarons_nbv_decision on macarons_amd Scene objects vs the golden the REFERENCE's own tester body produced on
    its Scene / Cell / Camera objects (make_golden.py: gen_decision): frustum mask, view states (OR-accumulated over the two
    poses), supervision occupancy, counters and out-of-field flags bit-exact; signed distances 1e-5 of the 550 fill depth;
    occupancy field points in identical order, harmonics 1e-5, probabilities 1e-4; the five neighbour gains 1e-4; the chosen
    neighbour."""
    from macarons_amd.utility import macarons_utils as mu
    g = golden("macarons_decision")
    m = _models(dev)
    surface, proxy = _decision_scenes(g, dev)
    H, W = int(g["hw"][0]), int(g["hw"][1])
    n = len(g["proxy"])
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98, sensor_range=40., min_occ_for_proxy_points=0.1, seq_len=2048,
                distance_factor_th=17., image_height=H, image_width=W, carving_tolerance=0.05)
    assert abs(3 * proxy.distance_between_proxy_points - float(g["dts"])) < 1e-12
    dmask = np.unpackbits(g["dmask"])[:2 * H * W].reshape(2, H, W).astype(bool)
    for c in range(2):
        cam = mu.SceneCamera(mu.camera_record(g["Mview"][c], g["Mfull"][c], g["ndc"], g["eyes"][c], params.sensor_range).to(dev),
                             T(g["eyes"][c:c + 1], dev), float(g["zfar"]))
        nrec = torch.stack([mu.camera_record(g[f"nMview_{c}"][k], g[f"nMfull_{c}"][k], g["ndc"], g["n_eyes"][c, k], params.sensor_range)
                            for k in range(5)]).to(dev)
        torch.manual_seed(5100 + c)
        with torch.no_grad():
            r = mu.macarons_nbv_decision(params, m, proxy, surface, cam, T(g["depth"][c], dev), T(dmask[c], dev), nrec,
                                         T(g["n_eyes"][c], dev), dev, samples=T(g[f"u_{c}"], dev), return_signed_distances=True)
        fov = np.unpackbits(g[f"fov_mask_{c}"])[:n].astype(bool)
        assert np.array_equal(r["fov_mask"].cpu().numpy(), fov), c
        # bilinear weights come from pixel coordinates up to W = 114 (fp32 ulp 8e-6) and blend depths with the 1.1 zfar = 550 fill
        assert np.abs(r["signed_distances"].cpu().numpy()[fov] - g[f"sgn_{c}"]).max() < 1e-5 * np.abs(g[f"sgn_{c}"]).max(), c
        assert np.array_equal(proxy.view_states.cpu().numpy().astype(np.uint8), np.unpackbits(g[f"view_states_{c}"], axis=-1)[:, :98]), c
        assert np.array_equal(proxy.proxy_supervision_occ.cpu().numpy()[:, 0].astype(np.uint8), g[f"sup_occ_{c}"]), c
        assert np.array_equal(proxy.out_of_field.cpu().numpy()[:, 0].astype(np.uint8), g[f"oof_{c}"]), c
        assert np.array_equal(proxy.proxy_n_inside_fov.cpu().numpy()[:, 0].astype(np.uint8), g[f"n_inside_{c}"]), c
        assert np.array_equal(proxy.proxy_n_behind_depth.cpu().numpy()[:, 0].astype(np.uint8), g[f"n_behind_{c}"]), c
        assert np.array_equal(r["X_world"].cpu().numpy(), g[f"X_world_{c}"]), c
        assert rel_err(r["view_harmonics"].cpu().numpy()[::5], g[f"vh_{c}"]) < 1e-5, c
        scale = np.abs(g[f"occ_{c}"]).max()
        assert np.abs(r["occ_probs"].cpu().numpy() - g[f"occ_{c}"]).max() < 1e-4 * scale, c
        assert np.abs(proxy.proxy_proba.cpu().numpy() - g[f"proxy_proba_{c}"]).max() < 1e-4 * scale, c
        assert rel_err(r["gains"].cpu().numpy(), g[f"gains_{c}"]) < 1e-4, c
        assert int(r["next_idx"]) == int(g[f"next_idx_{c}"]), c
    # the unfused state-update methods (Scene.update_proxy_view_states / _supervision_occ / _out_of_field + the signed-distance
    # op) give the same state as the fused pass: redo decision 0's update on a fresh scene
    from macarons_amd import ops
    _, fresh = _decision_scenes(g, dev)
    rec = mu.camera_record(g["Mview"][0], g["Mfull"][0], g["ndc"], g["eyes"][0], params.sensor_range).to(dev)
    fm = ops.points_in_fov(fresh.proxy_points, rec.view(1, 40))[0]
    sgn = ops.signed_distance_to_depth(fresh.proxy_points[fm].contiguous(), rec, T(g["depth"][0], dev), T(dmask[0], dev), 1.1 * float(g["zfar"]))
    assert np.abs(sgn.cpu().numpy() - g["sgn_0"]).max() < 1e-5 * np.abs(g["sgn_0"]).max()
    fresh.update_proxy_view_states(NS(X_cam=T(g["eyes"][0:1], dev)), fm, signed_distances=sgn)
    fresh.update_proxy_supervision_occ(fm, sgn, tol=params.carving_tolerance)
    fresh.update_proxy_out_of_field(fm)
    assert np.array_equal(fresh.view_states.cpu().numpy().astype(np.uint8), np.unpackbits(g["view_states_0"], axis=-1)[:, :98])
    assert np.array_equal(fresh.proxy_supervision_occ.cpu().numpy()[:, 0].astype(np.uint8), g["sup_occ_0"])
    assert np.array_equal(fresh.out_of_field.cpu().numpy()[:, 0].astype(np.uint8), g["oof_0"])


def test_macarons_decision_range_guard_is_deferred_and_falls_back(dev):
    """The fp16-split range check of a MACARONS decision is read once at the end (no stall inside the occupancy pass); with weights
    that overflow the fp16 range the decision is repeated on the full-range variant with the same hidden draws: it reports
    fallback_variant = 5 and equals, bit for bit, the decision of a model that runs on variant 5 from the start."""
    import ctypes
    from macarons_amd import _lib
    from macarons_amd.utility import macarons_utils as mu
    g = golden("macarons_decision")
    H, W = int(g["hw"][0]), int(g["hw"][1])
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98, sensor_range=40., min_occ_for_proxy_points=0.1, seq_len=2048,
                distance_factor_th=17., image_height=H, image_width=W, carving_tolerance=0.05)
    dmask = np.unpackbits(g["dmask"])[:2 * H * W].reshape(2, H, W).astype(bool)
    L = _lib.lib()

    def decide(force_variant):
        m = _models(dev)
        with torch.no_grad():
            for lin in (m.occupancy.linear1, m.occupancy.x_embedding.linear2):     # activations beyond 65504 in the head
                lin.weight.mul_(65536.); lin.bias.mul_(65536.)                      # (the recipe of test_range_guard_falls_back_...)
        surface, proxy = _decision_scenes(g, dev)
        cam = mu.SceneCamera(mu.camera_record(g["Mview"][0], g["Mfull"][0], g["ndc"], g["eyes"][0], params.sensor_range).to(dev),
                             T(g["eyes"][0:1], dev), float(g["zfar"]))
        nrec = torch.stack([mu.camera_record(g["nMview_0"][k], g["nMfull_0"][k], g["ndc"], g["n_eyes"][0, k], params.sensor_range)
                            for k in range(5)]).to(dev)
        v0 = L.mcr_get_local_pct_variant()
        if force_variant:
            L.mcr_set_local_pct_variant(ctypes.c_int(force_variant))
        try:
            torch.manual_seed(5100)
            with torch.no_grad():
                r = mu.macarons_nbv_decision(params, m, proxy, surface, cam, T(g["depth"][0], dev), T(dmask[0], dev), nrec,
                                             T(g["n_eyes"][0], dev), dev, samples=T(g["u_0"], dev))
            assert m.occupancy.range_guard == "sync"                    # restored (the default)
        finally:
            L.mcr_set_local_pct_variant(ctypes.c_int(v0))
        return r

    if L.mcr_get_local_pct_variant() != 6:
        pytest.skip("the range guard belongs to variant 6")
    a, b = decide(None), decide(5)
    assert a.get("fallback_variant") == 5 and "fallback_variant" not in b
    assert torch.equal(a["occ_probs"], b["occ_probs"]) and torch.equal(a["gains"], b["gains"]) and int(a["next_idx"]) == int(b["next_idx"])


def test_coverage_metrics_match_reference(dev):
    """Scene.scene_coverage / Scene.camera_coverage_gain (macarons_utils.py:2987-3056: fp64 nearest distance against epsilon, per
    cell / against the whole in-box partial cloud) vs the values the reference's methods returned on its own Scene objects."""
    from macarons_amd.utility.scene import Scene
    g = golden("macarons_decision")
    surface, _ = _decision_scenes(g, dev)
    x_min, x_max, grid = T(g["x_min"], dev), T(g["x_max"], dev), [int(v) for v in g["grid"]]
    rec = Scene(x_min, x_max, *grid, cell_capacity=500, cell_resolution=0.2, n_proxy_points=len(g["proxy"]), device=dev, feature_dim=1)
    torch.manual_seed(int(g["cov_seed"]))
    rec.fill_cells(T(g["cov_part"], dev), features=torch.zeros(len(g["cov_part"]), 1, device=dev))
    cov, n_gt = surface.scene_coverage(rec, surface_epsilon=0.25)
    assert n_gt == int(g["cov_n"]) and float(cov) == float(g["cov_value"])
    surface.set_all_features_to_value(0.)
    for c in surface.cells.values():
        c.cell_features[::2] = 1.
    gain = surface.camera_coverage_gain(T(g["cov_part2"], dev), surface_epsilon=0.2)
    assert float(gain) == float(g["cov_gain"])


def run_macarons_trajectory(dev, variant=None, strict=True):
    """strict=False (with `variant`): the exact / 1e-4 value assertions of the default numerics become entries of the returned report
    (per step: differing mask bits, value errors, the choice), for a variant with its own stated tolerance (tests/test_variant7_gpu.py);
    the replay itself -- the reference's poses, clouds and hidden draws -- is the same.
    TEN consecutive decisions with the state accumulating (BASELINE config 5: "10 trajectory steps ... achieved surface coverage vs
    reference"): the golden is the REFERENCE's tester body (testers/scene.py:284-454) driven pose after pose on its own Scene / Cell /
    Camera objects -- 3 x 2 x 3 grid, 24 000 proxy points, the surface scene empty at the start and fed by every depth map, 5-8
    neighbour poses per step, the camera MOVING to the chosen one (make_golden.py: gen_trajectory).  Replayed here through
    Scene.fill_cells / scene_coverage / compute_partial_point_cloud / macarons_nbv_decision with the hidden draws keyed by position
    (tests/golden/keyed_rng.py; a draw the reference did not make, or made with another size, raises).  Per step: achieved surface
    coverage EXACT, frustum mask / supervision occupancy / out-of-field flags / counters / view-state row sums exact, the number of
    field rows, occupancies 1e-4, view harmonics 1e-5, every neighbour's gain 1e-4, the SAME ten choices; at the end the whole
    view-state table exact and the stored probabilities at 1e-4."""
    import contextlib
    import keyed_rng as KR
    from macarons_amd import ops
    from macarons_amd.utility import macarons_utils as mu
    report = {"steps": []}
    from macarons_amd.utility.scene import Scene
    g = golden("macarons_trajectory")
    m = _models(dev)
    G = float(g["G"])
    n_steps, base = int(g["n_steps"]), int(g["base_seed"])
    H, W = int(g["hw"][0]), int(g["hw"][1])
    P = len(g["proxy"])
    x_min, x_max, grid = T(g["x_min"], dev), T(g["x_max"], dev), [int(v) for v in g["grid"]]
    F = lambda a: T(a.astype(np.float32) / G, dev)                              # grid coordinates stored as int16 multiples of 2^-6
    stages = ["part_gt", "covered_fill", "part", "surface_fill", "decision"]
    expect = KR.unpack_sizes(g["rng_sizes"], g["rng_off"], n_steps, stages)
    expect[(-1, "gt_fill")] = [int(v) for v in g["gt_fill_sizes"]]
    mk = lambda cap, res, sc=1.: Scene(x_min, x_max, *grid, cell_capacity=cap, cell_resolution=res, n_proxy_points=P, device=dev,
                                       feature_dim=1, score_threshold=sc)
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98, sensor_range=float(g["sensor_range"]), min_occ_for_proxy_points=0.1,
                seq_len=2048, distance_factor_th=17., image_height=H, image_width=W, carving_tolerance=0.05)
    fixes = {}
    for (s_, k_, j_), v_ in zip(g["u_fix_rows"].tolist(), g["u_fix_vals"].tolist()):
        fixes.setdefault((s_, k_), {})[j_] = v_
    kr = KR.KeyedRandperm(base, expect=expect)
    zeros = lambda n: torch.zeros(n, 1, device=dev)
    with kr.installed(), torch.no_grad():
        kr.at(-1, "gt_fill")
        gt_scene, covered, surface, proxy = mk(3000, 0.15), mk(1500, 0.2), mk(500, 0.2), mk(100000, 1e-4, 0.95)
        gt_scene.fill_cells(F(g["gt"]), features=zeros(len(g["gt"])))
        proxy.initialize_proxy_points()
        proxy.proxy_points = F(g["proxy"])
        assert abs(3 * proxy.distance_between_proxy_points - float(g["dts"])) < 1e-12
        for step in range(n_steps):
            depth = T(g[f"depth_{step}"], dev)
            dmask = T(np.unpackbits(g[f"dmask_{step}"])[:H * W].reshape(H, W).astype(bool), dev)
            dcam = mu.depth_camera_record(g[f"Mfull_{step}"], float(g["P"][2, 2]), float(g["P"][3, 2]))
            snapped = lambda p: {tuple(r_) for r_ in np.round(p.cpu().numpy() * G).astype(np.int64).tolist()}
            # ---- ground-truth partial cloud -> covered scene -> achieved coverage (testers/scene.py:318-336)
            for stage, fill, scene, key in (("part_gt", "covered_fill", covered, "part_gt"), ("part", "surface_fill", surface, "part")):
                kr.at(step, stage)
                part = mu.compute_partial_point_cloud(depth.view(1, H, W, 1), dmask.view(1, H, W, 1), dcam, float(g["gf"]),
                                                      fov_range=params.sensor_range)
                want = {tuple(r_) for r_ in g[f"{key}_{step}"].astype(np.int64).tolist()}
                got = snapped(part)                      # the harness's sensor quantisation: the 2^-6 grid (cell faces excepted)
                assert len(part) == int(g[f"part_raw_n_{step}"][0 if key == "part_gt" else 1])
                assert len(got & want) >= 0.985 * len(want), (step, key, len(got & want), len(want))
                kr.at(step, fill)
                scene.fill_cells(F(g[f"{key}_{step}"]), features=zeros(len(g[f"{key}_{step}"])))   # (the reference's own snapped cloud: the state stays exact)
                if key == "part_gt":
                    cov, n_gt = gt_scene.scene_coverage(covered, surface_epsilon=float(g["eps_cov"]))
                    assert n_gt == int(g["cov_n"]) and float(cov) == float(g["coverage"][step]), (step, float(cov), float(g["coverage"][step]))
            assert np.array_equal(np.array([len(c.cell_pts) for _, c in sorted(surface.cells.items())]), g[f"surface_n_{step}"]), step
            # ---- the decision (:391-454)
            kr.at(step, "decision")
            eye = g[f"eye_{step}"]
            cam = mu.SceneCamera(mu.camera_record(g[f"Mview_{step}"], g[f"Mfull_{step}"], g["ndc"], eye, params.sensor_range).to(dev),
                                 T(eye[None], dev), float(g["zfar"]))
            n_eyes, nb = g[f"n_eyes_{step}"], g[f"nb_{step}"]
            K = len(n_eyes)
            nrec = torch.stack([mu.camera_record(g[f"nMview_{step}"][k], g[f"nMfull_{step}"][k], g["ndc"], n_eyes[k], params.sensor_range)
                                for k in range(K)]).to(dev)
            u = torch.stack([KR.keyed_uniforms(base, step, int(a) * 8 + int(e)).view(-1) for a, e in nb.tolist()])
            for k, (a, e) in enumerate(nb.tolist()):
                for j_, v_ in fixes.get((step, int(a) * 8 + int(e)), {}).items():
                    u[k, j_] = v_
            with (ops.variant(variant) if variant else contextlib.nullcontext()):
                r = mu.macarons_nbv_decision(params, m, proxy, surface, cam, depth, dmask, nrec, T(n_eyes, dev), dev, samples=u.to(dev))
            assert "fallback_variant" not in r                       # (the range guard of both networks stayed quiet)
            bits = lambda t_: np.packbits(t_.cpu().numpy().reshape(-1).astype(np.uint8))
            if not strict:
                nbits = lambda t_, ref: int(np.unpackbits(bits(t_) ^ ref).sum())
                occ_ref = g[f"occ_{step}"]
                gains, gref = r["gains"].cpu().numpy().reshape(-1), g[f"gains_{step}"].reshape(-1)
                order = np.argsort(-gref)
                report["steps"].append({
                    "fov_mask_bits": nbits(r["fov_mask"], g[f"fov_mask_{step}"]), "sup_occ_bits": nbits(proxy.proxy_supervision_occ, g[f"sup_occ_{step}"]),
                    "oof_bits": nbits(proxy.out_of_field, g[f"oof_{step}"]),
                    "vs_rowsum_diff": int((proxy.view_states.sum(-1).cpu().numpy().astype(np.uint8) != g[f"vs_rowsum_{step}"]).sum()),
                    "field_rows": int(r["X_world"].shape[0]), "field_rows_ref": int(g[f"field_n_{step}"]),
                    "occ_rel_err": (float(np.abs(r["occ_probs"].cpu().numpy()[::7, 0] - occ_ref).max() / np.abs(occ_ref).max())
                                    if r["X_world"].shape[0] == int(g[f"field_n_{step}"]) else None),
                    "gains_rel_err": rel_err(gains, gref), "next_idx": int(r["next_idx"]), "next_idx_ref": int(g["next_idx"][step]),
                    "ref_gain_margin_rel": float((gref[order[0]] - gref[order[1]]) / np.abs(gref).max()) if len(gref) > 1 else None})
                continue
            assert np.array_equal(bits(r["fov_mask"]), g[f"fov_mask_{step}"]), step
            assert np.array_equal(bits(proxy.proxy_supervision_occ), g[f"sup_occ_{step}"]), step
            assert np.array_equal(bits(proxy.out_of_field), g[f"oof_{step}"]), step
            assert np.array_equal(proxy.view_states.sum(-1).cpu().numpy().astype(np.uint8), g[f"vs_rowsum_{step}"]), step
            assert np.array_equal(proxy.proxy_n_inside_fov.cpu().numpy()[:, 0].astype(np.uint8), g[f"n_inside_{step}"]), step
            assert np.array_equal(proxy.proxy_n_behind_depth.cpu().numpy()[:, 0].astype(np.uint8), g[f"n_behind_{step}"]), step
            assert r["X_world"].shape[0] == int(g[f"field_n_{step}"]), step
            occ_ref = g[f"occ_{step}"]
            assert np.abs(r["occ_probs"].cpu().numpy()[::7, 0] - occ_ref).max() < 1e-4 * np.abs(occ_ref).max(), step
            assert rel_err(r["view_harmonics"].cpu().numpy()[::37], g[f"vh_{step}"]) < 1e-5, step
            assert rel_err(r["gains"].cpu().numpy(), g[f"gains_{step}"]) < 1e-4, (step, r["gains"].cpu().numpy(), g[f"gains_{step}"])
            assert int(r["next_idx"]) == int(g["next_idx"][step]), step
    if not strict:
        report["view_state_bits_final"] = int(np.unpackbits(np.packbits(proxy.view_states.cpu().numpy().astype(np.uint8), axis=-1) ^ g["view_states_final"]).sum())
        report["proxy_proba_final_rel_err"] = float(np.abs(proxy.proxy_proba.cpu().numpy()[:, 0] - g["proxy_proba_final"]).max()
                                                    / np.abs(g["proxy_proba_final"]).max())
        report["draws_missing"] = [str(k_) for k_ in expect if len(expect[k_]) != len(kr.sizes().get(k_, []))]
        return report
    assert np.array_equal(np.packbits(proxy.view_states.cpu().numpy().astype(np.uint8), axis=-1), g["view_states_final"])
    assert np.abs(proxy.proxy_proba.cpu().numpy()[:, 0] - g["proxy_proba_final"]).max() < 1e-4 * np.abs(g["proxy_proba_final"]).max()
    assert [k_ for k_ in expect if len(expect[k_]) != len(kr.sizes().get(k_, []))] == []       # every draw the reference made was made
    return report


def test_macarons_trajectory_matches_reference(dev):
    """The ten-decision trajectory golden on the default numerics: every exact / 1e-4 assertion of run_macarons_trajectory."""
    run_macarons_trajectory(dev)
