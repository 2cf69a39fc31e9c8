"""Parity of the glue kernels (view state, view harmonics, proxy sampling, frustum mask, n-camera gains)."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_err
from oracle import view_state as V, scorer

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_view_state_and_harmonics_golden(dev):
    from macarons_amd.utility import scone_utils as su
    g = golden("view_sampler")
    base, h_polar, h_azim = su.get_all_harmonics_under_degree(8, 7, 14, dev)
    assert np.abs(base.cpu().numpy() - g["base"]).max() < 1e-5        # setup-time table (torch trig/pow on device)
    assert np.array_equal(h_polar.cpu().numpy(), g["h_polar"]) and np.array_equal(h_azim.cpu().numpy(), g["h_azim"])
    vs = su.compute_view_state(T(g["pts"], dev), T(g["X_view"], dev), 7, 14).cpu().numpy()
    ref = np.unpackbits(g["view_state"], axis=-1)[..., :98].astype(np.float32)
    bad = np.argwhere((vs != ref).any(-1))
    margin = V.bin_boundary_margin(g["pts"], g["X_view"], 7, 14)
    # bit-exact except for rays within 2e-6 rad of a binning boundary (asin/acos differ by an ulp between libms)
    for b in bad:
        assert margin[tuple(b)].min() < 2e-6
    assert len(bad) <= 2
    vh = su.compute_view_harmonics(T(ref, dev), T(g["base"], dev), T(g["h_polar"], dev), T(g["h_azim"], dev), 7, 14)
    assert rel_err(vh.cpu().numpy(), g["view_harmonics"]) < 1e-5


def test_view_state_vs_oracle_large(dev):
    from macarons_amd import ops
    rng = np.random.default_rng(2)
    pts = rng.uniform(-.5, .5, (1, 100_000, 3)).astype(np.float32)
    Xv = rng.standard_normal((9, 3)).astype(np.float32)
    Xv = (1.5 * Xv / np.linalg.norm(Xv, axis=1, keepdims=True)).astype(np.float32)
    vs = ops.view_state(T(pts, dev), T(Xv, dev), 7, 14).cpu().numpy()
    ref = V.compute_view_state(pts, Xv, 7, 14)
    bad = np.argwhere((vs != ref).any(-1))
    margin = V.bin_boundary_margin(pts, Xv, 7, 14)
    assert len(bad) < 20
    for b in bad:
        assert margin[tuple(b)].min() < 2e-6
    assert np.all((vs == 0) | (vs == 1)) and np.all(vs.sum(-1) >= 1) and np.all(vs.sum(-1) <= 9)


def test_sampler_golden_and_properties(dev):
    from macarons_amd.utility import scone_utils as su
    g = golden("view_sampler")
    res, resh, inv = su.sample_proxy_points(T(g["s_X"], dev), T(g["s_preds"], dev), T(g["s_vh"], dev), 2048, 0.1,
                                            return_index=True, samples=T(g["s_u"], dev))
    res, resh, inv = res.cpu().numpy(), resh.cpu().numpy(), inv.cpu().numpy()
    # vs the reference run: same picks sample by sample, up to uniforms that sit on a CDF step
    assert V.sampler_tie_aware_match(res, inv, g["s_res"], g["s_inv"], g["s_X"], g["s_preds"], 0.1)
    # vs the oracle's exact-CDF convention on other data, incl. everything below / above the threshold
    rng = np.random.default_rng(8)
    for P, n in ((257, 64), (100_000, 2048), (5, 4096)):
        X = rng.uniform(-.5, .5, (P, 3)).astype(np.float32)
        preds = rng.uniform(0.0, 1.0, (P, 1)).astype(np.float32)
        vh = rng.standard_normal((P, 64)).astype(np.float32)
        u = rng.uniform(0, 1, n).astype(np.float32)
        u[:3] = [0.0, 0.99999994, 0.5]
        r, h, i = su.sample_proxy_points(T(X, dev), T(preds, dev), T(vh, dev), n, 0.1, return_index=True, samples=T(u, dev))
        ro, ho, io, _ = V.sample_proxy_points(X, preds, vh, u, 0.1, exact=True)
        assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(h.cpu().numpy(), ho) and np.array_equal(i.cpu().numpy(), io)
        assert np.all(r.cpu().numpy()[:, 3] > 0.1)


def test_points_in_fov(dev):
    from macarons_amd import ops
    rng = np.random.default_rng(3)
    P, C = 100_003, 5
    pts = rng.uniform(-30, 30, (P, 3)).astype(np.float32)
    cams = np.zeros((C, 40), np.float32)
    for c in range(C):
        R = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
        Tt = rng.uniform(-5, 5, 3).astype(np.float32)
        Mv = np.eye(4, dtype=np.float32); Mv[:3, :3] = R; Mv[3, :3] = Tt
        f = 1.0 / np.tan(np.deg2rad(60) / 2)
        zn, zf = 1.0, 100.0
        K = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, zf / (zf - zn), 1], [0, 0, -zf * zn / (zf - zn), 0]], np.float32)
        cams[c, :16] = Mv.reshape(-1)
        cams[c, 16:32] = (Mv @ K).astype(np.float32).reshape(-1)
        cams[c, 32:36] = [456 / 256 - 2 * 455 / 255, 456 / 256, -1.0, 1.0]          # macarons_utils.py:1929-1938
        cams[c, 36:39] = (-Tt @ R.T)
        cams[c, 39] = 0.0 if c == 0 else 40.0
    mask = ops.points_in_fov(T(pts, dev), T(cams, dev)).cpu().numpy()
    F = np.float32
    for c in range(C):
        Mv, Mp = cams[c, :16].reshape(4, 4), cams[c, 16:32].reshape(4, 4)
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        lin = lambda M, j: ((x * M[0, j] + y * M[1, j]) + z * M[2, j]) + M[3, j]
        zv, px, py, pw = lin(Mv, 2), lin(Mp, 0), lin(Mp, 1), lin(Mp, 3)
        with np.errstate(divide="ignore", invalid="ignore"):
            nx, ny = px / pw, py / pw
        m = (nx >= cams[c, 32]) & (nx <= cams[c, 33]) & (ny >= cams[c, 34]) & (ny <= cams[c, 35]) & (zv > 0)
        if cams[c, 39] > 0:
            d = pts - cams[c, 36:39]
            m &= np.sqrt(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(F)) < cams[c, 39]
        assert np.array_equal(mask[c], m)                                           # bit-exact mask
        assert 0 < m.sum() < P


def test_coverage_gain_multiple(dev):
    from macarons_amd.networks import SconeVis
    d = golden("scorer_b2_n500_c7")
    m = SconeVis().to(dev)
    g2, i2 = m.compute_coverage_gain_multiple(T(d["pts"], dev), T(d["harmonics"], dev), T(d["cams"], dev), 2)
    assert np.array_equal(i2.numpy(), d["multi2_idx"]) and rel_err(g2.cpu().numpy(), d["multi2"]) < 1e-4
    g3, i3 = m.compute_coverage_gain_multiple(T(d["pts"][:, :128], dev), T(d["harmonics"][:, :128], dev),
                                              T(d["cams"][:, :4], dev), 3)
    assert np.array_equal(i3.numpy(), d["multi3_idx"]) and rel_err(g3.cpu().numpy(), d["multi3"]) < 1e-4
    with pytest.raises(NameError):
        m.compute_coverage_gain_multiple(T(d["pts"], dev), T(d["harmonics"], dev), T(d["cams"], dev), 4)


def test_macarons_regime_scoring(dev):
    """Batched per-neighbour-camera scoring (predict_coverage_gain_for_single_camera) vs the oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import weights
    from macarons_amd.networks import SconeVis
    from macarons_amd.utility import macarons_utils as mu
    from oracle import macarons_regime as MR
    g = golden("macarons_regime")
    fac = mu.get_distance_factor_threshold(T(g["pts"], dev), T(g["cam"], dev), 17.).cpu().numpy()
    assert np.abs(fac - g["factor"]).max() < 1e-6 and np.abs(MR.distance_factor_threshold(g["pts"], g["cam"]) - g["factor"]).max() < 1e-6
    vis = SconeVis()
    sdv = weights.make_state_dict(weights.shapes_of(vis), 1)
    vis.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()})
    vis = vis.to(dev).eval()
    rng = np.random.default_rng(12)
    P, K = 20000, 4
    X = rng.uniform(-20, 20, (P, 3)).astype(np.float32)
    vh = (rng.standard_normal((P, 64)) * .3).astype(np.float32)
    occ = rng.uniform(0, 1, (P, 1)).astype(np.float32)
    recs, Mpred, cams_w = [], [], []
    f = 1.0 / np.tan(np.deg2rad(60) / 2)
    Kp = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, 100 / 99, 1], [0, 0, -100 / 99, 0]], np.float32)
    for k in range(K):
        R = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
        Tt = rng.uniform(-3, 3, 3).astype(np.float32)
        Mv = np.eye(4, dtype=np.float32); Mv[:3, :3] = R; Mv[3, :3] = Tt
        c = (-Tt @ R.T).astype(np.float32)
        recs.append(mu.camera_record(Mv, (Mv @ Kp).astype(np.float32), [456 / 256 - 2 * 455 / 255, 456 / 256, -1.0, 1.0], c,
                                     fov_range=0.0 if k == 3 else 45.0).numpy())
        Mpred.append(Mv); cams_w.append(c)
    recs[3][33] = recs[3][32] - 1.0            # camera 3: empty frustum (max_ndc_x < min_ndc_x) -> gain 0
    recs, Mpred, cams_w = np.stack(recs), np.stack(Mpred), np.stack(cams_w)
    u = rng.uniform(0, 1, (K, 2048)).astype(np.float32)
    with torch.no_grad():
        gains = mu.predict_coverage_gain_for_cameras(vis, T(X, dev), T(vh, dev), T(occ, dev), T(recs, dev), T(cams_w, dev),
                                                     T(Mpred, dev), 57.0, samples=T(u, dev)).cpu().numpy()
    ref = np.array([MR.coverage_gain_for_camera(sdv, X, vh, occ, recs[k], cams_w[k], Mpred[k], 57.0, u[k]) for k in range(K)])
    assert gains[3] == 0.0 and ref[3] == 0.0
    assert rel_err(gains, ref) < 1e-4


def test_scene_side_kernels(dev):
    """K11 segmented fp64 nearest distance (Cell.fill / coverage tests) and K12 depth unprojection."""
    from macarons_amd import ops
    from oracle import scene as S
    rng = np.random.default_rng(21)
    na = [0, 700, 1, 1300, 257]
    nb = [5, 900, 0, 2000, 513]
    A = rng.uniform(0, 10, (sum(na), 3)).astype(np.float32)
    B = rng.uniform(0, 10, (sum(nb), 3)).astype(np.float32)
    ao, bo = np.concatenate([[0], np.cumsum(na)]).astype(np.int64), np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
    d = ops.min_dist_segmented(T(A, dev), T(ao, dev), T(B, dev), T(bo, dev)).cpu().numpy()
    for s in range(len(na)):
        ref = S.min_dist(A[ao[s]:ao[s + 1]], B[bo[s]:bo[s + 1]])
        got = d[ao[s]:ao[s + 1]]
        assert np.array_equal(np.isinf(ref), np.isinf(got))
        ok = ~np.isinf(ref)
        assert np.abs(got[ok] - ref[ok]).max(initial=0) < 1e-12
        # the decisions the reference takes on it (strict > resolution, heaviside(eps - d) i.e. d < eps) are identical
        for eps in (0.05, 0.3, 1.0):
            assert np.array_equal(got > eps, ref > eps) and np.array_equal(got < eps, ref < eps)
    # depth unprojection: round trip through an explicit perspective camera
    H, W = 24, 40
    f = 1.0 / np.tan(np.deg2rad(60) / 2)
    zn, zf = 1.0, 100.0
    K = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, zf / (zf - zn), 1], [0, 0, -zf * zn / (zf - zn), 0]], np.float64)
    R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    Mv = np.eye(4); Mv[:3, :3] = R; Mv[3, :3] = rng.uniform(-2, 2, 3)
    Minv = np.linalg.inv(Mv @ K)
    depth = rng.uniform(2, 50, (1, H, W)).astype(np.float32)
    cam = np.concatenate([Minv.reshape(-1), [K[2, 2], K[3, 2]]]).astype(np.float32)[None]
    w = ops.unproject_depth(T(depth, dev), T(cam, dev)).cpu().numpy()[0]
    ref = S.unproject_depth(depth[0], cam[0, :16].reshape(4, 4), cam[0, 16], cam[0, 17])
    assert np.abs(w - ref).max() < 2e-3 * np.abs(ref).max()
    # and the geometry closes: re-projecting the world points gives back the pixel ndc and the depth
    ph = np.concatenate([w.astype(np.float64), np.ones((H * W, 1))], 1)
    view = ph @ Mv
    assert np.abs(view[:, 2] - depth.reshape(-1)).max() < 1e-2
    nx, ny = S.ndc_tabs(H, W)
    proj = ph @ (Mv @ K)
    assert np.abs(proj[:, 0] / proj[:, 3] - nx.reshape(-1)).max() < 1e-3 and np.abs(proj[:, 1] / proj[:, 3] - ny.reshape(-1)).max() < 1e-3


def test_best_record_and_merge(dev):
    """The arg-max exchange records of the camera-sharded decision == torch.max over the full gain vector
    (testers/shapenet.py:172), first occurrence on ties, for every way of cutting the cameras into shards."""
    from macarons_amd import ops
    rng = np.random.default_rng(12)
    for B, C in ((1, 200), (3, 7), (2, 1000), (5, 64)):
        g = rng.random((B, C)).astype(np.float32)
        g[:, C // 3] = g.max(axis=1)                         # an exact tie at a lower ...
        g[:, C // 3 + 2 if C // 3 + 2 < C else C - 1] = g[:, C // 3]   # ... and a higher index
        G = torch.from_numpy(g).to(dev)
        ref = torch.max(G, dim=1)
        for world in (1, 2, 3, 8):
            if world > C:
                continue
            recs = []
            for r in range(world):
                q, rem = divmod(C, world)
                lo = r * q + min(r, rem)
                hi = lo + q + (1 if r < rem else 0)
                recs.append(ops.best_record(G[:, lo:hi].contiguous(), lo))
            vals, idx = ops.best_merge(torch.stack(recs, 0).contiguous())
            assert torch.equal(vals, ref.values) and torch.equal(idx, ref.indices), (B, C, world)


def test_filter_proxy_points(dev):
    """mcr_filter_proxy_points: the reference golden (bit-exact mask), a larger random case against the oracle, and the host
    mirror's two ways of receiving the view cameras."""
    from macarons_amd import ops
    from macarons_amd.utility import scone_utils as su
    from oracle import view_state as V
    g = golden("filter_proxy")
    ref_mask = np.unpackbits(g["mask"])[:len(g["X"])].astype(bool)
    mask, bounds = ops.filter_proxy_mask(T(g["X"], dev), T(g["pc"], dev), T(g["proj"], dev), float(g["tol"]))
    assert np.array_equal(mask.cpu().numpy(), ref_mask)
    _, ob = V.filter_proxy_points(g["proj"], g["X"], g["pc"], float(g["tol"]))
    assert np.array_equal(bounds.cpu().numpy(), ob)
    # larger, 7 views, with points behind some cameras (w < 0 flips the projection like the reference's division does)
    rng = np.random.default_rng(5)
    P, M, nv = 100_003, 10_240, 7
    X = rng.uniform(-0.6, 0.6, (P, 3)).astype(np.float32)
    d = rng.standard_normal((M, 3))
    pc = (d / np.linalg.norm(d, axis=1, keepdims=True) * [0.35, 0.25, 0.3]).astype(np.float32)
    proj = np.zeros((nv, 4, 4), np.float32)
    f = 1.0 / np.tan(np.deg2rad(60) / 2)
    K = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, 1000 / 999, 1], [0, 0, -1000 / 999, 0]], np.float32)
    for v in range(nv):
        R = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
        Mv = np.eye(4, dtype=np.float32); Mv[:3, :3] = R; Mv[3, :3] = [0, 0, 1.5]
        proj[v] = Mv @ K
    om, _ = V.filter_proxy_points(proj, X, pc, 0.01)
    Xf, m = su.filter_proxy_points(T(proj, dev), T(X, dev), T(pc, dev), filter_tol=0.01)
    assert np.array_equal(m.cpu().numpy(), om) and 0 < om.sum() < P
    assert torch.equal(Xf, T(X, dev)[m])

    class Cams:                                              # what a PyTorch3D camera batch offers
        def get_full_projection_transform(self):
            class Tr:
                def get_matrix(s):
                    return torch.from_numpy(proj)
            return Tr()
    _, m2 = su.filter_proxy_points(Cams(), T(X, dev), T(pc, dev), filter_tol=0.01)
    assert torch.equal(m2, m)
    with pytest.raises(NameError):
        su.filter_proxy_points(T(proj, dev), T(X[None], dev), T(pc, dev))


def test_move_view_state_to_view_space(dev):
    """Host mirror + mcr_gather_columns against the reference golden: the mirror bins with the same torch CPU ops the
    reference used, so all seven cameras (incl. the axis-aligned ones on bin boundaries) must match bit for bit."""
    from macarons_amd import ops
    from macarons_amd.utility import scone_utils as su
    g = golden("view_space")
    vs = np.unpackbits(g["view_state"], axis=-1)[..., :98].astype(np.float32)
    for c in range(int(g["n_cam"])):
        rot = np.unpackbits(g[f"rot_{c}"], axis=-1)[..., :98].astype(np.float32)
        xinv = torch.from_numpy(g[f"xinv_{c}"])

        class Cam:                                           # the three calls the reference makes on a PyTorch3D camera
            def get_world_to_view_transform(self):
                class Tr:
                    def inverse(s):
                        return s

                    def transform_points(s, p):
                        return xinv.to(p.device)             # moved grid + centre (captured from the reference run)
                return Tr()

            def get_camera_center(self):
                return torch.zeros(1, 3, device=dev)
        out = su.move_view_state_to_view_space(T(vs, dev), Cam(), 7, 14)
        assert np.array_equal(out.cpu().numpy(), rot), c
    # handed the rotation itself: same result wherever the direction is not on a bin boundary
    from oracle import view_state as V
    out = su.move_view_state_to_view_space(T(vs, dev), torch.from_numpy(g["R"][0]), 7, 14).cpu().numpy()
    rot0 = np.unpackbits(g["rot_0"], axis=-1)[..., :98].astype(np.float32)
    safe = V.view_space_bin_margin(g["xinv_0"], 7, 14) > 3e-6
    assert np.array_equal(out[..., safe], rot0[..., safe])
    idx = torch.randperm(98)
    x = torch.randn(5, 1000, 98, device=dev)
    assert torch.equal(ops.gather_columns(x, idx), x[..., idx.to(dev)])


@pytest.mark.gpu
def test_nbv_decide_follows_torch_max_and_the_empty_sample_rule():
    """mcr_nbv_decide = where(n_unique < 1, NaN, gains) -> torch.max over the cameras (testers/shapenet.py:172: first maximum, a
    NaN wins) -> index -1 for the empty clouds, plus the read-back record (range flag, indices, maxima)."""
    from macarons_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, C = 7, 200
    gains = torch.rand(B, C, generator=g)
    gains[1, 17] = gains[1, 150] = 2.0                       # a tie: the first one
    gains[2, 40] = float("nan"); gains[2, 90] = float("nan")  # NaN beats every number, the first NaN
    gains[3] = float("-inf")                                 # still a valid index (0)
    gains[5, 199] = 3.0                                      # the last column
    nu = torch.tensor([5, 1, 9, 2, 0, 2048, 0], dtype=torch.int32)
    ref = torch.where(nu.view(-1, 1) < 1, torch.full_like(gains, float("nan")), gains)
    best = torch.max(ref, dim=1)
    ref_idx = torch.where(nu < 1, torch.full_like(best.indices, -1), best.indices)
    flag = torch.tensor([1], dtype=torch.int32, device=dev)
    gd = gains.to(dev)
    mx, idx, rec = ops.nbv_decide(gd, nu.to(dev), flag)
    assert torch.equal(idx.cpu(), ref_idx)
    assert torch.equal(torch.nan_to_num(mx.cpu(), nan=-7.0), torch.nan_to_num(best.values, nan=-7.0))
    assert torch.equal(torch.nan_to_num(gd.cpu(), nan=-7.0), torch.nan_to_num(ref, nan=-7.0))          # the NaN rows, in place
    rec = rec.cpu()
    assert rec.dtype == torch.float64 and rec.numel() == 1 + 2 * B and rec[0] == 1.0
    assert torch.equal(rec[1:1 + B].to(torch.int64), ref_idx)
    assert torch.equal(torch.nan_to_num(rec[1 + B:].to(torch.float32), nan=-7.0), torch.nan_to_num(best.values, nan=-7.0))
    mx2, idx2, rec2 = ops.nbv_decide(gains[:1].contiguous().to(dev))                                       # no counts, no flag
    assert int(idx2) == int(torch.argmax(gains[0])) and rec2[0] == 0.0


def test_scene_grid_kernels_match_the_torch_expressions(dev):
    """mcr_cell_keys / mcr_key_histogram / mcr_admit_keys (the fused bookkeeping of Scene.fill_cells and of the occupancy field's cell
    lookup) against the torch expressions they replace (get_cells_for_each_pt's floor rule, the closed scene box, Cell.fill's strict
    box masks, per-cell counts, the fp64 admission compare): bit-exact on points inside, outside, on cell faces and on the scene's
    boundary, for two grids (one with more than 1023 cells: the histogram's device-op form)."""
    from macarons_amd import ops
    from macarons_amd.utility.scene import Scene
    rng = np.random.default_rng(12)
    for grid, lo_c, hi_c in (((3, 2, 3), [-12., -6., -12.], [12., 6., 12.]), ((11, 10, 12), [-1.3, -2.0, 0.5], [3.1, 2.2, 4.0])):
        x_min, x_max = torch.tensor(lo_c, device=dev), torch.tensor(hi_c, device=dev)
        sc = Scene(x_min, x_max, *grid, cell_capacity=1000, cell_resolution=0.05, n_proxy_points=10, device=dev, feature_dim=1)
        n_cells = grid[0] * grid[1] * grid[2]
        ext = np.array(hi_c) - np.array(lo_c)
        pts = rng.uniform(-0.1, 1.1, (20000, 3)) * ext + np.array(lo_c)                            # some outside the box
        step = ext / np.array(grid)
        faces = np.array(lo_c) + rng.integers(0, np.array(grid) + 1, (3000, 3)) * step             # exactly on cell faces / the boundary
        mix = rng.uniform(0, 1, (3000, 3)) * ext + np.array(lo_c)
        on = rng.random((3000, 3)) < 0.4
        pts = torch.from_numpy(np.concatenate([pts, np.where(on, faces, mix)]).astype(np.float32)).to(dev)
        valid = torch.from_numpy(rng.random(len(pts)) < 0.8).to(dev)
        cells, lo, hi = sc._cell_table()
        # the torch expressions of Scene.fill_cells (round 3 form)
        cid = (sc.get_cells_for_each_pt(pts) * sc._consts(dev)["lin"]).sum(-1)
        ok = ((pts >= x_min) & (pts <= x_max)).all(-1)
        ok = ok & (torch.max(pts - hi[cid], dim=-1)[0] < 0.) & (torch.min(pts - lo[cid], dim=-1)[0] > 0.) & valid
        key_ref = torch.where(ok, cid, torch.full_like(cid, n_cells))
        key = ops.cell_keys(pts, sc._consts(dev)["gc"], grid, lo, hi, valid)
        assert torch.equal(key.long(), key_ref)
        assert torch.equal(sc.linear_cell_ids(pts).long(), cid)
        assert 0 < int((key < n_cells).sum()) < len(pts)
        counts, offsets = ops.key_histogram(key, n_cells)
        ref = torch.bincount(key_ref, minlength=n_cells + 1)
        assert torch.equal(counts, ref) and torch.equal(offsets[1:], torch.cumsum(ref, 0)) and int(offsets[0]) == 0
        order = torch.sort(key, stable=True).indices
        key_s = key[order].contiguous()
        d = torch.from_numpy(rng.uniform(0, 0.1, len(pts))).to(dev)
        d[::7] = 0.05                                                                               # exactly the resolution: not admitted (strict)
        for n_min in (0, 3):
            k2 = ops.admit_keys(d, key_s, counts, 0.05, n_min, n_cells)
            adm = (d > 0.05) & (key_s < n_cells) & (counts[key_s.long()] > n_min)
            assert torch.equal(k2.long(), torch.where(adm, key_s.long(), torch.full_like(key_s.long(), n_cells)))


# ---- the fused scene-side kernels of one MACARONS decision (scene.hip) ------------------------------------------------------------------
@pytest.mark.parametrize("N,nk,runs", [(1, 3, False), (63, 1, False), (2048, 72, False), (2049, 72, True), (100_000, 72, False),
                                       (230_000, 72, True), (50_000, 1023, False), (4097, 300, True)])
def test_group_by_key_is_the_stable_sort(dev, N, nk, runs):
    """mcr_group_by_key == torch.sort(key, stable=True).indices, torch.bincount and its exclusive prefix sums: random keys, long runs of
    one key (cell ids of points in cloud order), keys outside 0 .. nk (counted as nk), sizes around the 2048-row tile."""
    from macarons_amd import ops
    g = torch.Generator().manual_seed(N + nk)
    key = torch.randint(0, nk + 1, (N,), generator=g, dtype=torch.int32)
    if runs:
        key = torch.repeat_interleave(torch.randint(0, nk + 1, (N // 37 + 1,), generator=g, dtype=torch.int32), 37)[:N].contiguous()
        key[::101] = -5                                      # not a key: the last group
        key[7::203] = nk + 9
    order, counts, offsets = ops.group_by_key(key.to(dev), nk)
    eff = torch.where((key < 0) | (key > nk), torch.full_like(key, nk), key).long()
    ref = torch.sort(eff, stable=True).indices
    cnt = torch.bincount(eff, minlength=nk + 1)
    assert torch.equal(order.cpu().long(), ref)
    assert torch.equal(counts.cpu(), cnt)
    assert torch.equal(offsets.cpu(), torch.cat((torch.zeros(1, dtype=torch.int64), torch.cumsum(cnt, 0))))


def test_uniform_rows_are_the_per_camera_torch_rand_draws(dev):
    """ops.uniform_rows(K, S) == K consecutive torch.rand(S, 1, device=...) calls, bit for bit, and leaves the device generator where those
    calls would leave it (upstream draws the sampling uniforms camera by camera, scone_utils.py:1052): pins torch's Philox indexing, the
    offset step per call and rocRAND's uniform map -- if a torch upgrade changes any of them this fails, it does not drift."""
    from macarons_amd import ops
    for seed, K, S, warm in ((1234, 30, 2048, 0), (7, 1, 2048, 3), (99, 5, 16, 1), (3, 4, 5000, 2)):
        torch.manual_seed(seed)
        for _ in range(warm):
            torch.rand(17, device=dev)                                     # the generator is not at offset 0
        want = torch.cat([torch.rand(S, 1, device=dev) for _ in range(K)], 1).t().contiguous()
        after_want = torch.rand(8, device=dev)
        torch.manual_seed(seed)
        for _ in range(warm):
            torch.rand(17, device=dev)
        got = ops.uniform_rows(K, S, dev)
        after_got = torch.rand(8, device=dev)
        assert torch.equal(got, want), (seed, K, S, float((got - want).abs().max()))
        assert torch.equal(after_got, after_want)


def test_view_harmonics_rows_and_camera_boxes_and_indexed_gain(dev):
    from macarons_amd import ops
    from macarons_amd.utility import scone_utils as su
    g = torch.Generator().manual_seed(11)
    # view harmonics of selected rows with a bin permutation == gather_columns + the [98] x [98, 64] product
    vs = (torch.rand(5000, 98, generator=g) < 0.06).float().to(dev)
    rows = torch.randperm(5000, generator=g)[:1777].to(torch.int32).to(dev)
    perm = torch.randperm(98, generator=g)
    base, h_polar, h_azim = su.get_all_harmonics_under_degree(8, 7, 14, dev)
    m = su._view_harmonics_matrix(base, h_polar, 7, 14)                     # [64, 98]
    got = ops.view_harmonics_rows(vs, rows, perm.to(torch.int32).to(dev), m.t().contiguous())
    want = (vs[rows.long()][:, perm.to(dev)].double() @ m.t().double()).float()
    assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 2e-6
    assert torch.equal(ops.view_harmonics_rows(vs, None, None, m.t().contiguous())[rows.long()],
                       ops.view_harmonics_rows(vs, rows, None, m.t().contiguous()))
    # prediction boxes (macarons_utils.py:1631-1660)
    K, S = 7, 300
    res = torch.randn(K, S, 4, generator=g).to(dev)
    nu = torch.tensor([300, 1, 0, 17, 299, 64, 65], dtype=torch.int32, device=dev)
    Mv = torch.randn(K, 4, 4, generator=g).to(dev)
    cam = torch.randn(K, 3, generator=g).to(dev)
    center, cam_v = ops.camera_boxes(res, nu, Mv, cam, 0.25)
    for k in range(K):
        n = int(nu[k])
        cw = (res[k, :n, :3].amax(0) + res[k, :n, :3].amin(0)) / 2. if n else torch.zeros(3, device=dev)
        c = (torch.cat((cw, torch.ones(1, device=dev))).double() @ Mv[k].double())[:3]
        assert torch.allclose(center[k].double(), c, rtol=1e-5, atol=1e-6), k
        cv = ((torch.cat((cam[k], torch.ones(1, device=dev))).double() @ Mv[k].double())[:3] - c) * 0.25
        assert torch.allclose(cam_v[k].double(), cv, rtol=1e-5, atol=1e-5), k
    # gains through the inverse map == gather the Monte-Carlo duplicates, then mcr_macarons_gain  (bit for bit: same sum, same order)
    vis_u = torch.rand(K, S, generator=g).to(dev)
    inv = torch.stack([torch.randint(0, max(int(n), 1), (S,), generator=g) for n in nu.tolist()]).to(dev)
    vol = (torch.rand(K, generator=g) * 100).to(dev)
    for th, smooth in ((1.3, False), (0.9, True)):
        got = ops.macarons_gain_indexed(vis_u, res, inv, nu, cam, vol, th, smooth)
        vis_mc = torch.gather(vis_u, 1, inv).contiguous()
        world = torch.gather(res, 1, inv[..., None].expand(-1, -1, 4)).contiguous()
        want = ops.macarons_gain_(vis_mc, world, cam, vol, th, smooth)
        want = torch.where(nu > 0, want, torch.zeros_like(want))
        assert torch.equal(got, want)
