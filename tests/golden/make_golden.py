#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REAL reference (this container only).

    python tests/golden/make_golden.py [group ...]      # groups: scorer sh knn view sampler blocks vis occ e2e

The reference Python lives read-only at /root/reference and never travels; what is committed is
data only (inputs + the reference's outputs, as small .npz files) together with this script.
Weights for the network goldens are NOT stored: they come from tests/golden/weights.py (our own
deterministic generator) and are loaded into the reference modules through their state_dict, so the
same weights can be rebuilt on the GPU box without the reference.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_import  # noqa: E402

ref = _ref_import.load_reference()
import torch  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def t(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


# --------------------------------------------------------------------------------------------
def cameras_on_sphere(n_elev, n_azim, radius=1.5):
    """The reference camera lattice (scone_utils.py:765-771) evaluated through the reference's
    own get_cartesian_coords."""
    elev = [-90. + (i + 1) / (n_elev + 1) * 180. for i in range(n_elev) for j in range(n_azim)]
    azim = [360. * j / n_azim for i in range(n_elev) for j in range(n_azim)]
    X = ref["CustomGeometry"].get_cartesian_coords(r=torch.full((len(elev), 1), radius),
                                                   elev=torch.Tensor(elev).view(-1, 1),
                                                   azim=torch.Tensor(azim).view(-1, 1), in_degrees=True)
    return X.numpy()


def gen_scorer():
    """G1: SconeVis.compute_coverage_gain / compute_visibilities / compute_coverage_gain_multiple."""
    SconeVis = ref["SconeVis"].SconeVis
    for tag, (B, N, C, sigma, seed) in {
        "scorer_b1_n2048_c20": (1, 2048, 20, 0.5, 11),
        "scorer_b2_n500_c7": (2, 500, 7, 1.5, 12),
    }.items():
        rng = np.random.default_rng(seed)
        pts = np.concatenate([rng.uniform(-0.5, 0.5, (B, N, 3)), rng.uniform(0.1, 1.0, (B, N, 1))], -1).astype(np.float32)
        harm = (rng.standard_normal((B, N, 64)) * sigma).astype(np.float32)
        if C == 20:
            cams = np.broadcast_to(cameras_on_sphere(4, 5), (B, C, 3)).astype(np.float32).copy()
        else:
            cams = (rng.standard_normal((B, C, 3))).astype(np.float32)
            cams = (1.5 * cams / np.linalg.norm(cams, axis=-1, keepdims=True)).astype(np.float32)
        out = {}
        for use_sigmoid in (True, False):
            m = SconeVis(use_sigmoid=use_sigmoid)
            sfx = "sig" if use_sigmoid else "relu"
            with torch.no_grad():
                out[f"gain32_{sfx}"] = m.compute_coverage_gain(t(pts), t(harm), t(cams)).numpy()
                out[f"vis32_{sfx}"] = m.compute_visibilities(t(pts), t(harm), t(cams)).numpy()
                out[f"gain64_{sfx}"] = m.compute_coverage_gain(t(pts, torch.float64), t(harm, torch.float64),
                                                               t(cams, torch.float64)).numpy()
                out[f"vis64_{sfx}"] = m.compute_visibilities(t(pts, torch.float64), t(harm, torch.float64),
                                                             t(cams, torch.float64)).numpy()
        if C == 7:
            m = SconeVis(use_sigmoid=True)
            with torch.no_grad():
                g2, idx2 = m.compute_coverage_gain_multiple(t(pts), t(harm), t(cams), 2)
                g3, idx3 = m.compute_coverage_gain_multiple(t(pts[:, :128]), t(harm[:, :128]), t(cams[:, :4]), 3)
            out.update(multi2=g2.numpy(), multi2_idx=idx2.numpy(), multi3=g3.numpy(), multi3_idx=idx3.numpy())
        save(tag, pts=pts, harmonics=harm, cams=cams, **out)


def gen_sh():
    """G2: get_spherical_harmonics l=0..7 on an angle grid incl. poles and phi ~ 0, +-pi;
    get_spherical_coords / get_cartesian_coords on assorted rays incl. axis-aligned ones."""
    sh = ref["spherical_harmonics"]
    geo = ref["CustomGeometry"]
    th = np.concatenate([np.linspace(0, np.pi, 33), [1e-4, np.pi - 1e-4, 0.3]])
    ph = np.concatenate([np.linspace(-np.pi, np.pi, 41), [1e-4, -1e-4, 0.5]])
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    TH, PH = TH.reshape(-1), PH.reshape(-1)
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        sh.clear_spherical_harmonics_cache()
        theta, phi = t(TH, dt), t(PH, dt)
        z = torch.cat([sh.get_spherical_harmonics(l, theta, phi) for l in range(8)], dim=-1)
        res["Y_" + name] = z.numpy()
    rng = np.random.default_rng(5)
    rays = rng.standard_normal((512, 3))
    rays[:6] = [[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]]
    rays[6:10] = [[1e-4, 1, 0], [0, 2, 1e-4], [-1e-3, 0.5, -2], [0, 0.3, -1]]
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        r, e, a = geo.get_spherical_coords(t(rays, dt))
        res["r_" + name], res["elev_" + name], res["azim_" + name] = r.numpy(), e.numpy(), a.numpy()
        back = geo.get_cartesian_coords(r.view(-1, 1), e.view(-1, 1), a.view(-1, 1))
        res["cart_" + name] = back.numpy()
    save("sh_basis", theta=TH, phi=PH, rays=rays, **res)


def gen_knn():
    """G3: utils.get_knn_points (cdist + topk + knn_gather).  (a) inputs on a 2^-10 grid: d^2 is exact in fp32 in
    any formulation, so indices are well defined up to exact ties; (b) real-valued inputs for the tie-aware check."""
    get_knn_points = ref["utils"].get_knn_points
    rng = np.random.default_rng(21)
    Xg = (rng.integers(-512, 512, (1, 2000, 3)) / 1024.0).astype(np.float32)
    pcg = (rng.integers(-512, 512, (1, 5000, 3)) / 1024.0).astype(np.float32)
    pts, d, idx = get_knn_points(t(Xg), t(pcg), 16)
    Xr = rng.uniform(-0.5, 0.5, (2, 700, 3)).astype(np.float32)
    pcr = rng.uniform(-0.5, 0.5, (2, 1500, 3)).astype(np.float32)
    ptsr, dr, idxr = get_knn_points(t(Xr), t(pcr), 16)
    # tiny cloud: fewer than 25 points takes cdist's direct path
    Xs = rng.uniform(-0.5, 0.5, (1, 9, 3)).astype(np.float32)
    pcs = rng.uniform(-0.5, 0.5, (1, 20, 3)).astype(np.float32)
    ptss, ds, idxs = get_knn_points(t(Xs), t(pcs), 16)
    save("knn", Xg=Xg, pcg=pcg, idx_g=idx.numpy().astype(np.int32), dist_g=d.numpy(), pts_g=pts.numpy()[:, :50],
         Xr=Xr, pcr=pcr, idx_r=idxr.numpy().astype(np.int32), dist_r=dr.numpy(),
         Xs=Xs, pcs=pcs, idx_s=idxs.numpy().astype(np.int32), dist_s=ds.numpy())


def _load(module, seed):
    import weights
    sd = weights.make_state_dict(weights.shapes_of(module), seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return module.eval()


def gen_blocks():
    """G8: attention / Embedding / Encoder for the dimension sets of the hot path."""
    A = ref["Attention"]
    rng = np.random.default_rng(31)
    out = {}
    for tag, (E, qk, N, Bb) in {"vis": (256, 64, 130, 1), "occ": (128, 32, 16, 9)}.items():
        x = rng.standard_normal((Bb, N, E)).astype(np.float32)
        enc = _load(A.Encoder(seq_len=N, qk_dim=qk, embedding_dim=E, n_heads=4), 100 + E)
        with torch.no_grad():
            out[f"enc_{tag}_x"] = x
            out[f"enc_{tag}_y"] = enc(t(x)).numpy()
            q = rng.standard_normal((Bb, 4, N, qk // 4)).astype(np.float32)
            k = rng.standard_normal((Bb, 4, N, qk // 4)).astype(np.float32)
            v = rng.standard_normal((Bb, 4, N, E // 4)).astype(np.float32)
            out[f"att_{tag}_q"], out[f"att_{tag}_k"], out[f"att_{tag}_v"] = q, k, v
            out[f"att_{tag}_y"] = A.attention(t(q), t(k), t(v)).numpy()
    emb_v = _load(A.Embedding(4, 256, global_feature=True, concatenate_input=True), 7)
    emb_o = _load(A.Embedding(3, 128, global_feature=False, concatenate_input=True), 8)
    xv = rng.uniform(-0.5, 0.5, (2, 77, 4)).astype(np.float32)
    xo = rng.uniform(-0.2, 0.2, (11, 16, 3)).astype(np.float32)
    with torch.no_grad():
        out.update(emb_vis_x=xv, emb_vis_y=emb_v(t(xv)).numpy(), emb_occ_x=xo, emb_occ_y=emb_o(t(xo)).numpy())
    save("blocks", **out)


def gen_masked():
    """attention / Encoder / SconeVis.forward / PCTransformer.forward WITH a mask (Attention.py:24-27: masked_fill(mask == 0, -1e3)
    BEFORE the division by sqrt(d)): random [B,1,N,N] masks incl. a fully masked query row (which attends uniformly over the keys --
    the -1e3 rule, not -inf) and a fully masked key column, a [N,N] mask shared by the batch, for both dimension sets of the hot
    path; 130 tokens (MFMA kernel, ragged last tile), 16 tokens (neighbourhood kernel) and 520 tokens (key split; every 4th query
    row stored).  Inputs are fp16-representable (stored as fp16: half the bytes, exact in fp32)."""
    A = ref["Attention"]
    rng = np.random.default_rng(37)
    h = lambda a: np.asarray(a, np.float16)
    out = {}
    for tag, (E, qk, N, Bb) in {"vis": (256, 64, 130, 1), "occ": (128, 32, 16, 9), "long": (128, 32, 520, 1)}.items():
        q, k = h(rng.standard_normal((Bb, 4, N, qk // 4))), h(rng.standard_normal((Bb, 4, N, qk // 4)))
        v = h(rng.standard_normal((Bb, 4, N, E // 4)))
        mask = (rng.random((Bb, 1, N, N)) < 0.6)
        mask[0, 0, 3, :] = False                     # a query with every key masked
        mask[-1, 0, :, 5] = False                    # a key nobody may attend to
        mask[0, 0, 7, :] = True
        f = lambda a: t(a.astype(np.float32))
        with torch.no_grad():
            out[f"{tag}_q"], out[f"{tag}_k"], out[f"{tag}_v"], out[f"{tag}_mask"] = q, k, v, np.packbits(mask)
            att = A.attention(f(q), f(k), f(v), mask=torch.from_numpy(mask)).numpy()
            out[f"{tag}_att"] = att if tag != "long" else att[:, :, ::4]
            if tag != "long":
                x = h(rng.standard_normal((Bb, N, E)))
                enc = _load(A.Encoder(seq_len=N, qk_dim=qk, embedding_dim=E, n_heads=4), 100 + E)
                out[f"{tag}_x"] = x
                out[f"{tag}_enc"] = enc(f(x), mask=torch.from_numpy(mask)).numpy()
                if tag == "occ":
                    out[f"{tag}_att_shared"] = A.attention(f(q), f(k), f(v), mask=torch.from_numpy(mask[0, 0])).numpy()   # [N,N]: broadcast over batch and heads
    vis = _load(ref["SconeVis"].SconeVis(), 1)
    N = 150
    pts = h(np.concatenate([rng.uniform(-0.5, 0.5, (2, N, 3)), rng.uniform(0.1, 1, (2, N, 1))], -1))
    vh = h(rng.standard_normal((2, N, 64)) * 0.3)
    mask = (rng.random((2, 1, N, N)) < 0.7)
    mask[1, 0, 10, :] = False
    pct = _load(ref["SconeOcc"].PCTransformer(seq_len=N, pts_embedding_dim=128, feature_dim=512), 12)
    pc = h(rng.uniform(-0.4, 0.4, (2, N, 3)))
    f = lambda a: t(a.astype(np.float32))
    with torch.no_grad():
        out.update(sv_pts=pts, sv_vh=vh, sv_mask=np.packbits(mask), sv_y=vis(f(pts), mask=torch.from_numpy(mask), view_harmonics=f(vh)).numpy(),
                   pct_pc=pc, pct_y=pct(f(pc), mask=torch.from_numpy(mask)).numpy())
    save("blocks_masked", **out)


def gen_vis():
    """G5: SconeVis.forward with deterministic weights (tests/golden/weights.py, seed 1)."""
    m = _load(ref["SconeVis"].SconeVis(), 1)
    rng = np.random.default_rng(41)
    out = {}
    for N in (16, 333, 2048):
        pts = np.concatenate([rng.uniform(-0.5, 0.5, (1, N, 3)), rng.uniform(0.1, 1.0, (1, N, 1))], -1).astype(np.float32)
        vh = (rng.standard_normal((1, N, 64)) * 0.3).astype(np.float32)
        with torch.no_grad():
            y = m(t(pts), view_harmonics=t(vh)).numpy()
            y64 = m.double()(t(pts, torch.float64), view_harmonics=t(vh, torch.float64)).numpy()
            m.float()
        out[f"pts_{N}"], out[f"vh_{N}"], out[f"y_{N}"] = pts, vh, y
        if N != 2048:
            out[f"y64_{N}"] = y64.astype(np.float32)
        else:
            out["fp32_vs_fp64_maxabs_2048"] = np.float64(np.abs(y - y64).max())
    B2 = np.concatenate([rng.uniform(-0.5, 0.5, (3, 100, 3)), rng.uniform(0.1, 1.0, (3, 100, 1))], -1).astype(np.float32)
    vh2 = (rng.standard_normal((3, 100, 64)) * 0.3).astype(np.float32)
    with torch.no_grad():
        out.update(pts_b3=B2, vh_b3=vh2, y_b3=m(t(B2), view_harmonics=t(vh2)).numpy())
    save("scone_vis", **out)


def shell_cloud(rng, M):
    """Points on a unit-diagonal ellipsoid shell + small noise (SURVEY §8d synthetic surface cloud)."""
    d = rng.standard_normal((M, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (d * np.array([0.35, 0.25, 0.3]) + rng.standard_normal((M, 3)) * 0.002).astype(np.float32)


def gen_occ():
    """G6: SconeOcc.forward with deterministic weights (seed 2) and the three randperm draws CAPTURED."""
    m = _load(ref["SconeOcc"].SconeOcc(), 2)
    rng = np.random.default_rng(51)
    out = {}
    for tag, (M, Q) in {"m100_q17": (100, 17), "m1024_q300": (1024, 300), "m4096_q512": (4096, 512)}.items():
        pc = shell_cloud(rng, M)[None]
        x = rng.uniform(-0.5, 0.5, (1, Q, 3)).astype(np.float32)
        vh = (rng.standard_normal((1, Q, 64)) * 0.3).astype(np.float32)
        perms = []
        real = torch.randperm

        def capture(n, *a, **kw):
            p = real(n, *a, **kw)
            perms.append(p.numpy().copy())
            return p
        torch.randperm = capture
        try:
            torch.manual_seed(1000 + M)
            with torch.no_grad():
                y = m(t(pc), t(x), t(vh)).numpy()
        finally:
            torch.randperm = real
        ds = m.n_scale and int(np.power(M / (16 * 8), 1. / 2)) or 2
        ds = 2 if ds == 0 else ds
        cut = [perms[0][:2048], perms[1][:M // ds], perms[2][:(M // ds) // ds]]
        with torch.no_grad():
            gf = m.global_transformer(t(pc)[:, torch.from_numpy(cut[0])]).numpy()
        out.update({f"{tag}_pc": pc, f"{tag}_x": x, f"{tag}_vh": vh, f"{tag}_y": y, f"{tag}_gfeat": gf,
                    f"{tag}_perm0": cut[0].astype(np.int32), f"{tag}_perm1": cut[1].astype(np.int32),
                    f"{tag}_perm2": cut[2].astype(np.int32), f"{tag}_seed": np.int64(1000 + M)})
    save("scone_occ", **out)


def gen_view():
    """G4: get_all_harmonics_under_degree(8,7,14), compute_view_state, compute_view_harmonics;
    G7: sample_proxy_points with the uniforms captured."""
    import importlib
    su = importlib.import_module("macarons.utility.scone_utils")
    base, h_polar, h_azim = su.get_all_harmonics_under_degree(8, 7, 14, "cpu")
    rng = np.random.default_rng(61)
    Q = 1000
    pts = rng.uniform(-0.5, 0.5, (2, Q, 3)).astype(np.float32)
    pts[0, :4] = [[0, 0, 0], [0.1, 0, 0.1], [0, 0.2, 0], [-0.3, 0.1, 0]]
    X_view = cameras_on_sphere(4, 5)[[0, 3, 7, 12, 19]].astype(np.float32)
    X_view = np.concatenate([X_view, [[0, 1.5, 0], [0, 0, -1.5], [1.5, 0, 0]]]).astype(np.float32)   # axis-aligned views
    vs = su.compute_view_state(t(pts), t(X_view), 7, 14)
    vh = su.compute_view_harmonics(vs, base, h_polar, h_azim, 7, 14)
    out = dict(base=base.numpy(), h_polar=h_polar.numpy(), h_azim=h_azim.numpy(), pts=pts, X_view=X_view,
               view_state=np.packbits(vs.numpy().astype(np.uint8), axis=-1), view_harmonics=vh.numpy())
    # sampler
    P = 4000
    Xw = rng.uniform(-0.5, 0.5, (P, 3)).astype(np.float32)
    preds = rng.uniform(-0.1, 1.0, (P, 1)).astype(np.float32)
    vhp = (rng.standard_normal((P, 64)) * 0.3).astype(np.float32)
    drawn = []
    real = torch.rand

    def capture(*a, **kw):
        r = real(*a, **kw)
        drawn.append(r.numpy().copy())
        return r
    torch.rand = capture
    try:
        torch.manual_seed(77)
        res, resh, inv = su.sample_proxy_points(t(Xw), t(preds), t(vhp), 2048, 0.1, use_occ_to_sample=True, return_index=True)
    finally:
        torch.rand = real
    out.update(s_X=Xw, s_preds=preds, s_vh=vhp, s_u=drawn[0].reshape(-1), s_res=res.numpy(), s_resh=resh.numpy(),
               s_inv=inv.numpy().astype(np.int32))
    save("view_sampler", **out)


def gen_e2e():
    """G9: config-1 end-to-end NBV decision through the reference's own functions (testers/shapenet.py:126-172):
    chair-like shell cloud of 1024 points, 2048 proxy points, 1 past view, 20 candidate cameras.  Weights: seeds 2/1
    with occupancy linear3.bias shifted by +0.5 so untrained occupancies pass min_occ (SURVEY §8c gotcha)."""
    import importlib
    import weights
    su = importlib.import_module("macarons.utility.scone_utils")
    occ = _load(ref["SconeOcc"].SconeOcc(), 2)
    vis = _load(ref["SconeVis"].SconeVis(), 1)
    with torch.no_grad():
        occ.linear3.bias += 0.5
    rng = np.random.default_rng(71)
    pc = shell_cloud(rng, 1024)[None]
    X = rng.uniform(-0.5, 0.5, (1, 2048, 3)).astype(np.float32)
    X_cam = cameras_on_sphere(4, 5).astype(np.float32)
    X_view = X_cam[[6]]
    base, h_polar, h_azim = su.get_all_harmonics_under_degree(8, 7, 14, "cpu")
    perms, drawn = [], []
    real_perm, real_rand = torch.randperm, torch.rand

    def cap_perm(n, *a, **kw):
        p = real_perm(n, *a, **kw); perms.append(p.numpy().copy()); return p

    def cap_rand(*a, **kw):
        r = real_rand(*a, **kw); drawn.append(r.numpy().copy()); return r
    torch.randperm, torch.rand = cap_perm, cap_rand
    try:
        torch.manual_seed(123)
        with torch.no_grad():
            vs = su.compute_view_state(t(X), t(X_view), 7, 14)
            vh = su.compute_view_harmonics(vs, base, h_polar, h_azim, 7, 14)
            occ_prob = su.compute_occupancy_probability(occ, t(pc), t(X), vh, max_points_per_pass=300000).view(-1, 1)
            proxy, vh_s, sample_idx = su.sample_proxy_points(t(X)[0], occ_prob, vh.squeeze(0), n_sample=2048, min_occ=0.1,
                                                             use_occ_to_sample=True, return_index=True)
            harm = vis(proxy[None], view_harmonics=vh_s[None])
            proxy_mc = proxy[sample_idx][None]
            harm_mc = harm[0][sample_idx][None]
            gains = vis.compute_coverage_gain(proxy_mc, harm_mc, t(X_cam).view(1, -1, 3)).view(-1)
            max_gain, max_idx = torch.max(gains, dim=0)
    finally:
        torch.randperm, torch.rand = real_perm, real_rand
    M = 1024
    ds = int(np.power(M / (16 * 8), 1. / 2)) or 2
    cut = [perms[0][:2048], perms[1][:M // ds], perms[2][:(M // ds) // ds]]
    save("e2e_config1", pc=pc, X=X, X_view=X_view, X_cam=X_cam, occ=occ_prob.numpy(), n_unique=np.int64(proxy.shape[0]),
         gains=gains.numpy(), nbv_idx=np.int64(max_idx.item()), samples=drawn[0].reshape(-1), seed=np.int64(123),
         perm0=cut[0].astype(np.int32), perm1=cut[1].astype(np.int32), perm2=cut[2].astype(np.int32),
         occ_bias_shift=np.float32(0.5))


def gen_macarons():
    """get_distance_factor_threshold (macarons_utils.py:1768-1776) + the final product of
    predict_coverage_gain_for_single_camera (:1699-1704) on reference pieces."""
    import importlib
    mu = importlib.import_module("macarons.utility.macarons_utils")
    rng = np.random.default_rng(81)
    pts = rng.uniform(-40, 40, (3000, 3)).astype(np.float32)
    cam = np.array([[3.0, -2.0, 5.0]], np.float32)
    fac = mu.get_distance_factor_threshold(t(pts), t(cam), distance_th=17.)
    save("macarons_regime", pts=pts, cam=cam, factor=fac.numpy())



class _StandInCameras:
    """PyTorch3D is not installed in the build container.  The reference helpers that take a camera batch only use
    `.R.shape[0]`, `get_full_projection_transform().transform_points(p)` and (move_view_state_to_view_space)
    `get_world_to_view_transform().inverse().transform_points(p)` / `get_camera_center()`: this stand-in supplies them from
    explicit row-vector matrices, with Transform3d.transform_points as published ([x y z 1] M, divided by w) evaluated in
    the fixed order the HIP kernels and the oracle use.  The reference FUNCTION's own logic is what the golden pins."""

    class _T:
        def __init__(self, M, squeeze=False):
            self.M = M                                   # [n,4,4]
            self.squeeze = squeeze
            self._shape = None

        def _in_shape(self, p):
            return self._shape

        def transform_points(self, p):
            self._shape = tuple(p.shape)
            p = p.reshape(-1, 3)
            x, y, z = p[:, 0], p[:, 1], p[:, 2]
            cols = [((x[None] * self.M[:, 0, j, None] + y[None] * self.M[:, 1, j, None]) + z[None] * self.M[:, 2, j, None])
                    + self.M[:, 3, j, None] for j in range(4)]
            out = torch.stack([cols[0] / cols[3], cols[1] / cols[3], cols[2] / cols[3]], -1)        # [n,P,3]
            if self.squeeze and out.shape[0] == 1 and len(self._in_shape(p)) == 2:
                out = out[0]                        # Transform3d.transform_points: [P,3] in, batch of one -> [P,3] out
            return out

        def inverse(self):
            return _StandInCameras._T(torch.linalg.inv(self.M.double()).float(), self.squeeze)

    def __init__(self, R, T, P=None, squeeze=False, fov=60.0):
        self.R, self.T = R, T
        self.squeeze = squeeze
        self.fov = torch.tensor([fov])
        n = R.shape[0]
        self.Mv = torch.zeros(n, 4, 4)
        self.Mv[:, :3, :3] = R
        self.Mv[:, 3, :3] = T
        self.Mv[:, 3, 3] = 1.0
        self.P = P

    def get_world_to_view_transform(self):
        return self._T(self.Mv, self.squeeze)

    def get_full_projection_transform(self):
        return self._T(self.Mv @ self.P, self.squeeze)

    def unproject_points(self, xy_depth, scaled_depth_input=False):
        """FoVPerspectiveCameras.unproject_points as published (pytorch3d 0.6.2 cameras.py): depth -> scaled depth with the
        projection entries K[2,2], K[2,3] (= P[2,2], P[3,2] in the row-vector matrices used here), then the inverse of the
        full projection transform."""
        assert not scaled_depth_input
        f1 = self.P[:, 2, 2].reshape(-1, 1, 1)
        f2 = self.P[:, 3, 2].reshape(-1, 1, 1)
        sdepth = (f1 * xy_depth[..., 2:3] + f2) / xy_depth[..., 2:3]
        xy_sdepth = torch.cat((xy_depth[..., 0:2], sdepth), dim=-1)
        Minv = torch.linalg.inv((self.Mv @ self.P).double())
        p4 = torch.cat((xy_sdepth.double(), torch.ones_like(xy_sdepth[..., :1]).double()), -1) @ Minv
        return (p4[..., :3] / p4[..., 3:4]).float()

    def get_camera_center(self):
        return -torch.einsum("nj,nij->ni", self.T, self.R)            # C = -T R^T


def _look_at(eye):
    """pytorch3d look_at_view_transform(eye=eye, at=0, up=+Y) as published: z = normalize(at - eye), x = normalize(up x z),
    y = z x x;  R = [x y z] as columns, T = -R^T eye."""
    eye = np.asarray(eye, np.float64)
    z = -eye / np.linalg.norm(eye, axis=-1, keepdims=True)
    up = np.broadcast_to([0.0, 1.0, 0.0], eye.shape)
    x = np.cross(up, z)
    x /= np.linalg.norm(x, axis=-1, keepdims=True)
    y = np.cross(z, x)
    R = np.stack([x, y, z], -1)                                       # columns
    T = -np.einsum("nij,ni->nj", R, eye)
    return R.astype(np.float32), T.astype(np.float32)


def _fov_projection(fov_deg=60.0, znear=1.0, zfar=1000.0):
    """FoVPerspectiveCameras default projection (row-vector convention = transpose of the published K)."""
    s = 1.0 / np.tan(np.deg2rad(fov_deg) / 2)
    K = np.array([[s, 0, 0, 0], [0, s, 0, 0], [0, 0, zfar / (zfar - znear), 1], [0, 0, -zfar * znear / (zfar - znear), 0]], np.float32)
    return K


def gen_filter():
    """filter_proxy_points (scone_utils.py:1001-1027) with the tester's camera construction (testers/shapenet.py:117-122)."""
    import importlib
    su = importlib.import_module("macarons.utility.scone_utils")
    rng = np.random.default_rng(91)
    X_view = cameras_on_sphere(4, 5)[[2, 9, 16]].astype(np.float32)
    R, T = _look_at(X_view)
    cams = _StandInCameras(t(R), t(T), t(np.broadcast_to(_fov_projection(), (3, 4, 4)).copy()))
    d = rng.standard_normal((3000, 3))
    pc = (d / np.linalg.norm(d, axis=1, keepdims=True) * [0.3, 0.2, 0.25] + 0.002 * rng.standard_normal((3000, 3))).astype(np.float32)
    X = rng.uniform(-0.5, 0.5, (20000, 3)).astype(np.float32)
    Xf, mask = su.filter_proxy_points(cams, t(X), t(pc), filter_tol=0.01)
    proj = cams.get_full_projection_transform().M.numpy()
    save("filter_proxy", X=X, pc=pc, proj=proj, tol=np.float32(0.01), mask=np.packbits(mask.numpy()), n_keep=np.int64(Xf.shape[0]))


def gen_viewspace():
    """move_view_state_to_view_space (scone_utils.py:863-931) on stand-in cameras; the grid directions moved to view space
    (the input of the reference's get_spherical_coords call) are captured so the binning + gather are pinned exactly."""
    import importlib
    su = importlib.import_module("macarons.utility.scone_utils")
    rng = np.random.default_rng(95)
    eyes = np.concatenate([cameras_on_sphere(4, 5)[[1, 8, 13, 18]], [[0.3, 1.2, -0.8], [1.5, 0.0, 0.0], [0.0, 0.0, 1.5]]]).astype(np.float32)
    R, T = _look_at(eyes)
    vs = (rng.random((2, 300, 98)) < 0.15).astype(np.float32)
    out = dict(view_state=np.packbits(vs.astype(np.uint8), axis=-1), R=R, T=T)
    real = su.get_spherical_coords
    for c in range(len(eyes)):
        cam = _StandInCameras(t(R[c:c + 1]), t(T[c:c + 1]), t(_fov_projection()[None]))
        seen = []

        def capture(X):
            seen.append(X.numpy().copy())
            return real(X)
        su.get_spherical_coords = capture
        try:
            rot = su.move_view_state_to_view_space(t(vs), cam, 7, 14)
        finally:
            su.get_spherical_coords = real
        idx = None
        out[f"xinv_{c}"] = seen[0].astype(np.float32)
        out[f"rot_{c}"] = np.packbits(rot.numpy().astype(np.uint8), axis=-1)
    save("view_space", n_cam=np.int64(len(eyes)), **out)


# ---------------------------------------------------------------------------------------------------------------------
# Round 2: the MACARONS-regime rows pinned on the reference's own functions.
def _grid(a, step=1024.0):
    """Quantise to the 2^-10 grid: squared distances are then exact in fp32 in ANY formulation (|x|^2+|y|^2-2xy or
    (x-y)^2), so torch.cdist + topk of the reference and a direct-form kNN see identical values."""
    return (np.round(np.asarray(a, np.float64) * step) / step).astype(np.float32)


def _boundary_ties(X, pc, k=16, step=1024.0):
    """Mask of the queries whose k-th and (k+1)-th neighbour sit at the same (exact) squared distance."""
    Xi = np.round(X.astype(np.float64) * step).astype(np.int64)
    Pi = np.round(pc.astype(np.float64) * step).astype(np.int64)
    bad = np.zeros(len(Xi), bool)
    for lo in range(0, len(Xi), 2048):
        d = ((Xi[lo:lo + 2048, None, :] - Pi[None, :, :]) ** 2).sum(-1)
        part = np.partition(d, k, axis=1)[:, :k + 1]
        part.sort(axis=1)
        bad[lo:lo + 2048] = part[:, k - 1] == part[:, k]
    return bad


def _e2e(tag, M, Q, n_elev_cam, n_azim_cam, view_ids, seed, torch_seed):
    """One SCONE NBV decision through the reference's own functions (testers/shapenet.py:126-172) on grid-quantised clouds."""
    import importlib
    su = importlib.import_module("macarons.utility.scone_utils")
    occ = _load(ref["SconeOcc"].SconeOcc(), 2)
    vis = _load(ref["SconeVis"].SconeVis(), 1)
    with torch.no_grad():
        occ.linear3.bias += 0.5
    rng = np.random.default_rng(seed)
    ds = int(np.power(M / (16 * 8), 1. / 2)) or 2
    pc = np.unique(_grid(shell_cloud(rng, M + 64)), axis=0)
    rng.shuffle(pc)
    pc = pc[:M]
    assert pc.shape == (M, 3)
    # the three randperm draws SconeOcc.forward will make under this seed (checked below against the captured ones)
    torch.manual_seed(torch_seed)
    pp = [torch.randperm(M).numpy(), torch.randperm(M).numpy(), torch.randperm(M // ds).numpy()]
    pc1 = pc[pp[1][:M // ds]]
    pc2 = pc1[pp[2][:(M // ds) // ds]]
    # queries: redraw every query that has a k / k+1 distance tie in one of the three clouds, until none has: the
    # 16-neighbour SET of every query is then unique, whatever the tie order of topk
    X = _grid(rng.uniform(-0.5, 0.5, (Q, 3)))
    for it in range(100):
        bad = _boundary_ties(X, pc) | _boundary_ties(X, pc1) | _boundary_ties(X, pc2)
        _, first = np.unique(X, axis=0, return_index=True)
        dup = np.ones(Q, bool); dup[first] = False
        bad |= dup
        print(f"  {tag}: pass {it}: {int(bad.sum())} queries with boundary ties / duplicates")
        if not bad.any():
            break
        X[bad] = _grid(rng.uniform(-0.5, 0.5, (int(bad.sum()), 3)))
    else:
        raise RuntimeError("no tie-free query set found")
    perms, drawn = [], []
    real_perm, real_rand = torch.randperm, torch.rand

    def cap_perm(n, *a, **kw):
        p_ = real_perm(n, *a, **kw); perms.append(p_.numpy().copy()); return p_

    def cap_rand(*a, **kw):
        r_ = real_rand(*a, **kw); drawn.append(r_.numpy().copy()); return r_
    attempt = 0
    X_cam = cameras_on_sphere(n_elev_cam, n_azim_cam).astype(np.float32)
    X_view = X_cam[view_ids]
    base, h_polar, h_azim = su.get_all_harmonics_under_degree(8, 7, 14, "cpu")
    torch.randperm, torch.rand = cap_perm, cap_rand
    try:
        torch.manual_seed(torch_seed + attempt)
        with torch.no_grad():
            vs = su.compute_view_state(t(X[None]), t(X_view), 7, 14)
            vh = su.compute_view_harmonics(vs, base, h_polar, h_azim, 7, 14)
            occ_prob = su.compute_occupancy_probability(occ, t(pc[None]), t(X[None]), vh, max_points_per_pass=300000).view(-1, 1)
            proxy, vh_s, sample_idx = su.sample_proxy_points(t(X), occ_prob, vh.squeeze(0), n_sample=2048, min_occ=0.1,
                                                             use_occ_to_sample=True, return_index=True)
            harm = vis(proxy[None], view_harmonics=vh_s[None])
            proxy_mc = proxy[sample_idx][None]
            harm_mc = harm[0][sample_idx][None]
            gains = vis.compute_coverage_gain(proxy_mc, harm_mc, t(X_cam).view(1, -1, 3)).view(-1)
            max_gain, max_idx = torch.max(gains, dim=0)
    finally:
        torch.randperm, torch.rand = real_perm, real_rand
    assert all(np.array_equal(a, b) for a, b in zip(perms, pp)), "predicted permutations differ from the ones forward drew"
    cut = [perms[0][:2048], perms[1][:M // ds], perms[2][:(M // ds) // ds]]
    # unique original indices of the sampled points (the reference returns only the points): recover them by matching rows
    keep = np.nonzero(occ_prob.numpy()[:, 0] > np.float32(0.1))[0]
    save(tag, pc=pc[None], X=X[None], X_view=X_view, X_cam=X_cam, occ=occ_prob.numpy(), n_unique=np.int64(proxy.shape[0]),
         proxy=proxy.numpy(), sample_idx=sample_idx.numpy().astype(np.int32), harm_sum=harm.numpy().astype(np.float64).sum(1),
         gains=gains.numpy(), nbv_idx=np.int64(max_idx.item()), samples=drawn[0].reshape(-1), seed=np.int64(torch_seed + attempt),
         perm0=cut[0].astype(np.int32), perm1=cut[1].astype(np.int32), perm2=cut[2].astype(np.int32),
         occ_bias_shift=np.float32(0.5), n_kept=np.int64(len(keep)))


def gen_e2e_grid():
    """G9 on the 2^-10 grid (config 1: M=1024, Q=2048, C=20) and the config-2 shape (M=4096, Q=16384, C=100)."""
    _e2e("e2e_grid_config1", 1024, 2048, 4, 5, [6], 171, 1230)
    _e2e("e2e_grid_config2", 4096, 16384, 10, 10, [3, 47, 88], 172, 2230)


def _stub_renderer(H=256, W=456):
    from types import SimpleNamespace as NS
    return NS(rasterizer=NS(raster_settings=NS(image_size=(H, W))))


def _ref_camera(mu, H=256, W=456):  # noqa: E302
    """A REAL reference Camera (macarons_utils.py:1852): only its renderer is a stub that carries the image size, so the NDC
    tables and bounds (:1929-1938) are the reference's own arithmetic."""
    return mu.Camera(x_min=torch.tensor([-40., -40., -40.]), x_max=torch.tensor([40., 40., 40.]), pose_l=2, pose_w=2, pose_h=2,
                     pose_n_elev=3, pose_n_azim=4, n_interpolation_steps=1, zfar=500., renderer=_stub_renderer(H, W), device="cpu")


def _scene_cams(eyes, zfar=500.):
    R, T = _look_at(np.asarray(eyes, np.float32))
    P = np.broadcast_to(_fov_projection(60.0, 1.0, zfar), (len(R), 4, 4)).copy()
    return R, T, P


def gen_fov():
    """Camera.get_points_in_fov (macarons_utils.py:2400-2435) on a real Camera object + stand-in FoV cameras."""
    import importlib
    mu = importlib.import_module("macarons.utility.macarons_utils")
    cam = _ref_camera(mu)
    rng = np.random.default_rng(101)
    pts = rng.uniform(-40, 40, (30000, 3)).astype(np.float32)
    eyes = np.array([[30., 5., -20.], [0., 35., 1.], [-25., -10., 25.], [3., 2., 1.]], np.float32)
    R, T, P = _scene_cams(eyes)
    allc = _StandInCameras(t(R), t(T), t(P))
    out = dict(pts=pts, eyes=eyes, R=R, T=T, P=P, ndc=np.array([cam.min_ndc_x, cam.max_ndc_x, cam.min_ndc_y, cam.max_ndc_y], np.float32),
               Mview=allc.Mv.numpy(), Mfull=allc.get_full_projection_transform().M.numpy(), center=allc.get_camera_center().numpy(),
               ndc_x_tab=cam.ndc_x_tab.numpy()[::37, ::41], ndc_y_tab=cam.ndc_y_tab.numpy()[::37, ::41])
    for c in range(len(eyes)):
        fc = _StandInCameras(t(R[c:c + 1]), t(T[c:c + 1]), t(P[c:c + 1]), squeeze=True)
        for tag, rg in (("r40", 40.0), ("none", None)):
            sel, mask = cam.get_points_in_fov(t(pts), return_mask=True, fov_camera=fc, fov_range=rg)
            assert torch.equal(sel, t(pts)[mask])
            out[f"mask_{c}_{tag}"] = np.packbits(mask.numpy())
            out[f"n_{c}_{tag}"] = np.int64(mask.sum())
    save("fov_camera", **out)


def gen_distance():
    """get_distance_factor / _threshold / _smooth (macarons_utils.py:1741-1788)."""
    import importlib
    from types import SimpleNamespace as NS
    mu = importlib.import_module("macarons.utility.macarons_utils")
    rng = np.random.default_rng(103)
    pts = rng.uniform(-40, 40, (3000, 3)).astype(np.float32)
    cam = np.array([[3.0, -2.0, 5.0]], np.float32)
    params = NS(image_height=256, image_width=456)
    fc = NS(fov=torch.tensor([60.0]))
    res = 0.1                                                        # distance_th = f eps / pixel = 19.6 here
    save("distance_factors", pts=pts, cam=cam, cell_resolution=np.float32(res), fov=np.float32(60.0), hw=np.array([256, 456]),
         f_plain=mu.get_distance_factor(params, t(pts), t(cam), fc, res).numpy(),
         f_smooth=mu.get_distance_factor_smooth(params, t(pts), t(cam), fc, res).numpy(),
         f_th=mu.get_distance_factor_threshold(t(pts), t(cam), distance_th=17.).numpy())


def _ref_macarons(shift=0.5):
    M = importlib_macarons()
    occ = _load(ref["SconeOcc"].SconeOcc(), 2)
    vis = _load(ref["SconeVis"].SconeVis(), 1)
    with torch.no_grad():
        occ.linear3.bias += shift
    return M.Macarons(None, occ, vis).eval()


def importlib_macarons():
    import importlib
    return importlib.import_module("macarons.networks.Macarons")


def gen_macarons_wrapper():
    """Macarons.forward(mode=...) dispatch and compute_visibility_gains (Macarons.py:110-178)."""
    m = _ref_macarons()
    rng = np.random.default_rng(105)
    pc = shell_cloud(rng, 600)[None]
    x = rng.uniform(-0.5, 0.5, (1, 150, 3)).astype(np.float32)
    vh = (rng.standard_normal((1, 150, 64)) * 0.3).astype(np.float32)
    pts = np.concatenate([rng.uniform(-0.5, 0.5, (1, 200, 3)), rng.uniform(0.1, 1, (1, 200, 1))], -1).astype(np.float32)
    vh2 = (rng.standard_normal((1, 200, 64)) * 0.3).astype(np.float32)
    cams = rng.standard_normal((1, 5, 3)).astype(np.float32)
    cams = (1.5 * cams / np.linalg.norm(cams, axis=-1, keepdims=True)).astype(np.float32)
    perms = []
    real = torch.randperm

    def cap(n, *a, **kw):
        p_ = real(n, *a, **kw); perms.append(p_.numpy().copy()); return p_
    torch.randperm = cap
    try:
        torch.manual_seed(9)
        with torch.no_grad():
            o = m(mode='occupancy', partial_point_cloud=t(pc), proxy_points=t(x), view_harmonics=t(vh)).numpy()
    finally:
        torch.randperm = real
    with torch.no_grad():
        h = m(mode='visibility', proxy_points=t(pts), view_harmonics=t(vh2))
        g32 = m.compute_visibility_gains(pts=t(pts), harmonics=h, X_cam=t(cams)).numpy()
        g64 = m.double().compute_visibility_gains(pts=t(pts, torch.float64), harmonics=h.double(), X_cam=t(cams, torch.float64)).numpy()
    errs = {}
    for mode, kw in (("occupancy", dict(proxy_points=t(x))), ("visibility", dict(proxy_points=t(pts))), ("depth", {}), ("bogus", {})):
        try:
            m(mode=mode, **kw)
        except NameError as e:
            errs[mode] = str(e)
    M0, ds = 600, int(np.power(600 / (16 * 8), 1. / 2)) or 2
    save("macarons_wrapper", pc=pc, x=x, vh=vh, occ=o, perm0=perms[0][:2048].astype(np.int32), perm1=perms[1][:M0 // ds].astype(np.int32),
         perm2=perms[2][:(M0 // ds) // ds].astype(np.int32), pts=pts, vh2=vh2, cams=cams, harm=h.numpy(), gains32=g32, gains64=g64,
         err_occupancy=errs["occupancy"], err_visibility=errs["visibility"], err_depth=errs["depth"], err_bogus=errs["bogus"], seed=np.int64(9))


def gen_single_camera():
    """predict_coverage_gain_for_single_camera (macarons_utils.py:1580-1738): frustum -> occupancy filter -> sampling ->
    prediction-view space -> SconeVis -> per-point gains x distance factor -> mean x frustum volume; for several neighbour
    cameras incl. one whose frustum holds no occupied proxy point (the dummy branch, gain 0)."""
    import importlib
    from types import SimpleNamespace as NS
    mu = importlib.import_module("macarons.utility.macarons_utils")
    cam = _ref_camera(mu)
    m = _ref_macarons()
    rng = np.random.default_rng(107)
    P_ = 20000
    X_world = rng.uniform(-40, 40, (P_, 3)).astype(np.float32)
    # view harmonics as an elementwise (bit-reproducible) expression of three small stored factors: 20000 x 64 floats stay out of git
    vh_u, vh_v = (rng.standard_normal(P_) * 0.3).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    vh_w = (rng.standard_normal((16, 64)) * 0.2).astype(np.float32)
    vh = (vh_u[:, None] * vh_v[None, :] + vh_w[np.arange(P_) % 16]).astype(np.float32)
    occ = rng.uniform(-0.1, 1.0, (P_, 1)).astype(np.float32)
    occ[X_world[:, 0] < -20] = 0.0                                  # an empty slab: the last camera looks only at it
    eyes = np.array([[30., 5., -20.], [0., 35., 1.], [3., 2., 1.], [-39., 0., 0.]], np.float32)
    at = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [-80., 0., 0.]], np.float32)
    # look-at towards `at`: shift eyes so that _look_at (which looks at the origin) can be reused
    R, T = [], []
    for e, a in zip(eyes, at):
        r, _ = _look_at((e - a)[None])
        R.append(r[0]); T.append(-(r[0].T @ e))
    R, T = np.stack(R).astype(np.float32), np.stack(T).astype(np.float32)
    P = np.broadcast_to(_fov_projection(60.0, 1.0, 500.), (len(R), 4, 4)).copy()
    Rp, Tp, Pp = _scene_cams(np.array([[10., 20., -30.]], np.float32))
    pred = _StandInCameras(t(Rp), t(Tp), t(Pp), squeeze=True)
    proxy_scene = NS(x_min=torch.tensor([-40., -40., -40.]), x_max=torch.tensor([40., 40., 40.]))
    surface_scene = NS(cell_resolution=0.1)
    params = NS(sensor_range=60.0, min_occ_for_proxy_points=0.1, seq_len=2048, use_occ_to_sample_proxy_points=True, jz=False,
                ddp=False, distance_factor_th=17.0, k_for_knn=16, n_harmonics=64, image_height=256, image_width=456)
    allc = _StandInCameras(t(R), t(T), t(P))
    out = dict(X_world=X_world, vh_u=vh_u, vh_v=vh_v, vh_w=vh_w, occ=occ, eyes=eyes, R=R, T=T, P=P, Rp=Rp, Tp=Tp,
               Mview=allc.Mv.numpy(), Mfull=allc.get_full_projection_transform().M.numpy(), center=allc.get_camera_center().numpy(),
               Mpred=pred.Mv.numpy(),
               ndc=np.array([cam.min_ndc_x, cam.max_ndc_x, cam.min_ndc_y, cam.max_ndc_y], np.float32),
               box_diag=np.float32(torch.linalg.norm(proxy_scene.x_max - proxy_scene.x_min).item()), sensor_range=np.float32(60.0))
    real = torch.rand
    for c in range(len(eyes)):
        fc = _StandInCameras(t(R[c:c + 1]), t(T[c:c + 1]), t(P[c:c + 1]), squeeze=True)
        drawn = []

        def cap(*a, **kw):
            r_ = real(*a, **kw); drawn.append(r_.numpy().copy()); return r_
        torch.rand = cap
        try:
            torch.manual_seed(500 + c)
            with torch.no_grad():
                pw, vhs, vg, cg = mu.predict_coverage_gain_for_single_camera(
                    params, m, proxy_scene, surface_scene, t(X_world), t(vh), t(occ), cam, t(eyes[c:c + 1]), fc, prediction_camera=pred)
        finally:
            torch.rand = real
        out[f"gain_{c}"] = cg.numpy()
        out[f"vis_{c}"] = vg.numpy()
        out[f"world_{c}"] = pw.numpy()
        if drawn:
            # the same per-point gains from the reference's modules run in FLOAT64 on the SAME sampled set (the fp32 run's own
            # asin -> cos -> acos chain is off by up to 6e-4 near the poles): the unique sampled points (SconeVis is
            # permutation-equivariant, so their order is free), prediction-view normalisation, SconeVis, compute_visibility_gains,
            # distance factor -- all in double
            import copy
            md = copy.deepcopy(m).double()
            uq, inv = torch.unique(pw[0], dim=0, return_inverse=True)
            vh_u = torch.zeros(len(uq), 64, dtype=torch.float64)
            vh_u[inv] = vhs[0].double()
            Mp = pred.Mv[0].double()
            tf = lambda p: torch.cat((p, torch.ones(len(p), 1, dtype=torch.float64)), 1) @ Mp
            ptsd = uq.double().clone()
            ctr = tf(((ptsd[:, :3].max(0)[0] + ptsd[:, :3].min(0)[0]) / 2.).view(1, 3))[:, :3]
            diag = torch.linalg.norm(proxy_scene.x_max - proxy_scene.x_min).item()
            ptsd[:, :3] = (tf(ptsd[:, :3])[:, :3] - ctr) / diag
            Xc = ((tf(t(eyes[c:c + 1], torch.float64))[:, :3] - ctr) / diag)[None]
            with torch.no_grad():
                hd = md(mode='visibility', proxy_points=ptsd[None], view_harmonics=vh_u[None])
                vg64 = md.compute_visibility_gains(pts=ptsd[inv][None], harmonics=hd[0][inv][None], X_cam=Xc)
                fac = mu.get_distance_factor_threshold(pw[0, :, :3].double(), t(eyes[c:c + 1], torch.float64), distance_th=17.)
            out[f"vis64_{c}"] = (vg64 * fac.view(1, 1, -1)).numpy()
            print(f"    fp64 per-point run: max |fp32 - fp64| = {float((vg.double() - vg64 * fac.view(1, 1, -1)).abs().max()):.2e}")
        if drawn:
            out[f"u_{c}"] = drawn[0].reshape(-1)
        print(f"  camera {c}: gain {cg.numpy().ravel()}  sampled {pw.shape}")
    # the 'smooth' and None distance-factor branches on camera 0 (same uniforms)
    for tag, th in (("smooth", "smooth"), ("plain", None)):
        params.distance_factor_th = th
        fc = _StandInCameras(t(R[0:1]), t(T[0:1]), t(P[0:1]), squeeze=True)
        torch.manual_seed(500)
        with torch.no_grad():
            _, _, _, cg = mu.predict_coverage_gain_for_single_camera(params, m, proxy_scene, surface_scene, t(X_world), t(vh), t(occ), cam,
                                                                     t(eyes[0:1]), fc, prediction_camera=pred)
        out[f"gain_0_{tag}"] = cg.numpy()
    save("single_camera", **out)


def gen_cell():
    """Cell.fill (macarons_utils.py:2551-2577): bounding-box masks, fp64 admission test against the points already in the
    cell, random cap at capacity (randperm captured), two successive fills."""
    import importlib
    mu = importlib.import_module("macarons.utility.macarons_utils")
    rng = np.random.default_rng(109)
    center = torch.tensor([[1.0, -2.0, 0.5]])
    cell = mu.Cell(center=center, l=torch.tensor(4.0), w=torch.tensor(3.0), h=torch.tensor(2.0), capacity=400, resolution=0.12, device="cpu")
    out = dict(center=center.numpy(), lwh=np.array([4.0, 3.0, 2.0], np.float32), capacity=np.int64(cell.capacity), resolution=np.float64(cell.resolution))
    real = torch.randperm
    for i in range(3):
        pts = (rng.uniform(-1, 1, (1500, 3)) * [2.6, 2.0, 1.4] + center.numpy()).astype(np.float32)
        perms = []

        def cap(n, *a, **kw):
            p_ = real(n, *a, **kw); perms.append(p_.numpy().copy()); return p_
        torch.randperm = cap
        before = cell.cell_pts.numpy().copy()
        try:
            torch.manual_seed(40 + i)
            cell.fill(t(pts))
        finally:
            torch.randperm = real
        out[f"pts_{i}"], out[f"before_{i}"], out[f"after_{i}"], out[f"perm_{i}"] = pts, before, cell.cell_pts.numpy().copy(), perms[0].astype(np.int32)
    save("cell_fill", **out)


def gen_unproject():
    """utils.project_depth_back_to_3D (utils.py:1458-1487) and Camera.compute_partial_point_cloud (macarons_utils.py:2362-2398)
    on stand-in cameras whose unproject_points restates PyTorch3D's published algorithm."""
    import importlib
    mu = importlib.import_module("macarons.utility.macarons_utils")
    H, W = 32, 57
    cam = _ref_camera(mu, H, W)
    rng = np.random.default_rng(111)
    eyes = np.array([[30., 5., -20.], [-25., -10., 25.]], np.float32)
    R, T, P = _scene_cams(eyes)
    fc = _StandInCameras(t(R), t(T), t(P))
    depth = rng.uniform(2.0, 80.0, (2, H, W, 1)).astype(np.float32)
    depth[0, :3, :5] = -1.0                                           # background pixels of project_depth_back_to_3D (depth > -1 kept)
    depth[:, 10:12] = -1.0
    world = ref["utils"].project_depth_back_to_3D(t(depth), fc)
    mask = (rng.random((1, H, W, 1)) < 0.8)
    fc1 = _StandInCameras(t(R[:1]), t(T[:1]), t(P[:1]))
    d1 = np.abs(depth[:1]) + 1.0
    perms = []
    real = torch.randperm

    def cap(n, *a, **kw):
        p_ = real(n, *a, **kw); perms.append(p_.numpy().copy()); return p_
    torch.randperm = cap
    try:
        torch.manual_seed(77)
        part = cam.compute_partial_point_cloud(t(d1), torch.from_numpy(mask), fov_cameras=fc1, gathering_factor=0.25, fov_range=60.0)
    finally:
        torch.randperm = real
    save("unproject", depth=depth, eyes=eyes, R=R, T=T, P=P, Mfull=fc.get_full_projection_transform().M.numpy(), world=world.numpy(), d1=d1, mask=np.packbits(mask), part=part.numpy(),
         perm=perms[0].astype(np.int32), H=np.int64(H), W=np.int64(W))


def gen_formats():
    """On-disk formats read through the REFERENCE's own loaders from the files it ships:
    get_validation_optimal_sequences + get_optimal_sequence (scone_utils.py:639-646, 699-711) on
    data/ShapeNetCore.v1/validation_optimal_trajectories.pt, SceneDataset (CustomDataset.py:313-362) on data/scenes/liberty.
    Committed: a 6-object slice of the trajectories file re-saved in the same pickle schema, the liberty scene directory's two
    data files verbatim (data, 6.6 KB), and what the reference's loaders returned for them."""
    import importlib
    import json
    import shutil
    su = importlib.import_module("macarons.utility.scone_utils")
    cd = importlib.import_module("macarons.utility.CustomDataset")
    out_dir = os.path.join(HERE, "ref_data")
    os.makedirs(os.path.join(out_dir, "scenes", "liberty"), exist_ok=True)
    real_load = torch.load
    torch.load = lambda *a, **kw: real_load(*a, **{**kw, "weights_only": False})       # reference predates the weights_only default
    try:
        seqs = su.get_validation_optimal_sequences(False, "cpu")
        keys = sorted(seqs.keys())
        sub = {k: seqs[k] for k in keys[:3] + keys[-3:]}
        torch.save(sub, os.path.join(out_dir, "validation_optimal_trajectories_slice.pt"))
        exp = {}
        for k in sub:
            idx, cov = su.get_optimal_sequence(seqs, f"/x/ShapeNetCore.v1/03001627/{k}/model.obj", 4)
            exp[k] = {"idx": idx.tolist(), "coverage": [float(c) for c in cov]}
        for f in ("settings.json", "occupied_pose.pt"):
            shutil.copyfile(os.path.join(_ref_import.REFERENCE_ROOT, "data", "scenes", "liberty", f), os.path.join(out_dir, "scenes", "liberty", f))
        ds = cd.SceneDataset(os.path.join(_ref_import.REFERENCE_ROOT, "data", "scenes"), scene_names=["liberty"])
        item = ds[0]
    finally:
        torch.load = real_load
    pose = item["occupied_pose"]
    meta = {"n_objects_in_full_file": len(seqs), "keys": list(sub.keys()), "expected": exp,
            "scene": {"scene_name": item["scene_name"], "obj_name": item["obj_name"], "settings": item["settings"],
                      "X_idx_shape": list(pose["X_idx"].shape), "occupied_shape": list(pose["occupied"].shape),
                      "X_idx_dtype": str(pose["X_idx"].dtype), "occupied_dtype": str(pose["occupied"].dtype),
                      "n_occupied": int(torch.as_tensor(pose["occupied"]).sum()), "X_idx_first": torch.as_tensor(pose["X_idx"])[:5].tolist(),
                      "X_idx_last": torch.as_tensor(pose["X_idx"])[-1].tolist()}}
    with open(os.path.join(out_dir, "expected.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", out_dir, os.path.getsize(os.path.join(out_dir, "validation_optimal_trajectories_slice.pt")), "bytes slice")


def gen_occ_field():
    """compute_scene_occupancy_probability_field (macarons_utils.py:1395-1540) on REAL reference Scene / Cell objects (4 grid
    cells), a stand-in prediction camera and the reference Macarons wrapper: per cell the 27-neighbourhood surface cloud and the
    cell's proxy points go to the prediction camera's view space, normalised by 3 x the cell diagonal, view states are rotated,
    SconeOcc is evaluated (hidden randperm draws captured); then the out-of-field points are appended.  World points sit on a
    2^-6 grid and the proxy points are redrawn until no query has a k / k+1 neighbour tie in any of its three clouds."""
    import importlib
    from types import SimpleNamespace as NS
    mu = importlib.import_module("macarons.utility.macarons_utils")
    m = _ref_macarons()
    rng = np.random.default_rng(131)
    G = 64.0
    x_min, x_max = torch.tensor([-8., -4., -8.]), torch.tensor([8., 4., 8.])
    grid = (2, 1, 2)
    n_proxy = 3000

    def new_scene(capacity, resolution, feature_dim):
        return mu.Scene(x_min=x_min, x_max=x_max, grid_l=grid[0], grid_w=grid[1], grid_h=grid[2], cell_capacity=capacity,
                        cell_resolution=resolution, n_proxy_points=n_proxy, device="cpu", feature_dim=feature_dim)
    # surface: an ellipsoid shell through all four cells, on the grid
    d = rng.standard_normal((2600, 3))
    surf = np.unique(_grid(d / np.linalg.norm(d, axis=1, keepdims=True) * [5.5, 2.8, 5.0] + 0.05 * rng.standard_normal((2600, 3)), G), axis=0)
    rng.shuffle(surf)
    torch.manual_seed(4000)
    surface_scene = new_scene(500, 0.2, 0)
    surface_scene.fill_cells(t(surf))
    cell_pts = {k: c.cell_pts.numpy().copy() for k, c in surface_scene.cells.items()}
    print("  surface cells:", {k: len(v) for k, v in cell_pts.items()})
    def draw_proxy(n):
        q = _grid(rng.uniform(-1, 1, (n, 3)) * [7.9, 3.9, 7.9], G)
        q[q == 0] = 1.0 / G                       # Cell.fill's box tests are strict: a point ON a cell face belongs to no cell
        return q
    proxy = draw_proxy(n_proxy)
    in_fov = rng.random(n_proxy) < 0.7
    sup_occ = (rng.random((n_proxy, 1)) < 0.8).astype(np.float32)
    vstates = (rng.random((n_proxy, 98)) < 0.1).astype(np.float32)
    Rp, Tp, Pp = _scene_cams(np.array([[6., 9., -14.]], np.float32))
    pred = _StandInCameras(t(Rp), t(Tp), t(Pp), squeeze=True)
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98)

    def build_proxy_scene():
        ps = new_scene(100000, 1e-4, 1)
        ps.initialize_proxy_points()
        ps.proxy_points = t(proxy)
        ps.proxy_supervision_occ = t(sup_occ)
        ps.view_states = t(vstates)
        ps.out_of_field = t((~in_fov).astype(np.float32)).view(-1, 1)
        idx = ps.get_proxy_indices_from_mask(torch.from_numpy(in_fov))
        torch.manual_seed(4001)
        ps.fill_cells(t(proxy)[torch.from_numpy(in_fov)], features=idx.view(-1, 1).float())
        return ps
    seed = 4002
    for it in range(60):
        ps = build_proxy_scene()
        assert sum(len(c.cell_pts) for c in ps.cells.values()) == int(in_fov.sum()), "a proxy point was refused by Cell.fill"
        # replay the loop of the function to learn each cell's clouds and the randperm draws it will make
        occ_mask = (ps.proxy_supervision_occ > 0.)[..., 0]
        fov_mask = (ps.out_of_field < 1.)[..., 0]
        cells = ps.get_englobing_cells(ps.proxy_points[occ_mask * fov_mask])
        torch.manual_seed(seed)
        bad_idx = []
        for cell in cells:
            pcw = surface_scene.get_pt_cloud_from_cells(surface_scene.get_neighboring_cells(cell), return_features=False).numpy()
            _, ind = ps.get_pt_cloud_from_cells(cell, return_features=True)
            cmask = ps.get_proxy_mask_from_indices(ind) * occ_mask
            Xw = ps.proxy_points[cmask].numpy()
            gi = np.nonzero(cmask.numpy())[0]
            if not (pcw.shape[0] > 64 and len(Xw) > 0):
                continue
            M = len(pcw)
            ds = int(np.power(M / (16 * 8), 1. / 2)) or 2
            for lo in range(0, len(Xw), 20000):
                torch.randperm(M)                              # global down-sample (order of SconeOcc.forward)
                p1 = torch.randperm(M).numpy()[:M // ds]
                p2 = torch.randperm(M // ds).numpy()[:(M // ds) // ds]
                pc1 = pcw[p1]; pc2 = pc1[p2]
                Xc = Xw[lo:lo + 20000]
                bad = (_boundary_ties(Xc, pcw, 16, G) | _boundary_ties(Xc, pc1, 16, G) | _boundary_ties(Xc, pc2, 16, G))
                bad_idx += gi[lo:lo + 20000][bad].tolist()
        print(f"  occ_field: pass {it}: {len(bad_idx)} proxy points with boundary ties")
        if not bad_idx:
            break
        proxy[bad_idx] = draw_proxy(len(bad_idx))
    else:
        raise RuntimeError("no tie-free proxy set found")
    perms = []
    real = torch.randperm

    def cap(n, *a, **kw):
        p_ = real(n, *a, **kw); perms.append(p_.numpy().copy()); return p_
    proba_before = ps.proxy_proba.numpy().copy()
    torch.randperm = cap
    try:
        torch.manual_seed(seed)
        with torch.no_grad():
            Xw_out, vh_out, occ_out = mu.compute_scene_occupancy_probability_field(params, m, None, surface_scene, ps, "cpu",
                                                                                   prediction_camera=pred)
    finally:
        torch.randperm = real
    out = dict(x_min=x_min.numpy(), x_max=x_max.numpy(), grid=np.array(grid), surface=surf, n_surface_cells=np.int64(len(cell_pts)),
               proxy=proxy, in_fov=np.packbits(in_fov), sup_occ=sup_occ[:, 0].astype(np.uint8), view_states=np.packbits(vstates.astype(np.uint8), axis=-1),
               Mpred=pred.Mv.numpy(), Rp=Rp, X_world=Xw_out.numpy(), view_harmonics=vh_out.numpy(), occ_probs=occ_out.numpy(),
               proxy_proba=ps.proxy_proba.numpy(), proba_before=proba_before, seed=np.int64(seed), n_perms=np.int64(len(perms)))
    for i, (k, v) in enumerate(sorted(cell_pts.items())):
        out[f"cellkey_{i}"] = np.array(eval(k))
        out[f"cellpts_{i}"] = v
    for i, (k, c) in enumerate(sorted(ps.cells.items())):
        out[f"pcellkey_{i}"] = np.array(eval(k))
        out[f"pcellidx_{i}"] = c.cell_features.numpy()[:, 0].astype(np.int32)
    print(f"  occ_field: {len(Xw_out)} points out ({int(occ_out.shape[0])} probs), {len(perms)} randperm draws")
    save("occ_field", **out)


def _look_at_target(eyes, ats):
    """World->view rotations / translations of cameras at `eyes` looking at `ats` (up = +Y), pytorch3d's look_at as published."""
    R, T = [], []
    for e, a in zip(np.asarray(eyes, np.float32), np.asarray(ats, np.float32)):
        r, _ = _look_at((e - a)[None])
        R.append(r[0]); T.append(-(r[0].T @ e))
    return np.stack(R).astype(np.float32), np.stack(T).astype(np.float32)


def _ellipsoid_depth(ndc_x, ndc_y, eye, R, axes, fov_deg=60.0):
    """View-space depth of the ellipsoid sum((x/axes)^2) = 1 along the ray of every pixel (its NDC coordinates from the
    reference Camera's own tables), and the hit mask.  R = world->view rotation (columns = camera axes)."""
    s = 1.0 / np.tan(np.deg2rad(fov_deg) / 2)
    d_view = np.stack([ndc_x.astype(np.float64) / s, ndc_y.astype(np.float64) / s, np.ones_like(ndc_x, np.float64)], -1)
    d = d_view @ R.astype(np.float64).T
    o = np.asarray(eye, np.float64)
    ax = np.asarray(axes, np.float64)
    A = ((d / ax) ** 2).sum(-1)
    B = 2 * ((o / ax) * (d / ax)).sum(-1)
    C = ((o / ax) ** 2).sum() - 1
    disc = B * B - 4 * A * C
    hit = disc > 0
    tt = (-B - np.sqrt(np.where(hit, disc, 0.0))) / (2 * A)
    hit &= tt > 0
    return np.where(hit, tt, -1.0).astype(np.float32), hit


def gen_decision():
    """TWO consecutive next-best-view decisions of the MACARONS loop (testers/scene.py:391-454, everything between the depth
    network and the move to the next pose) driven on REAL reference Scene / Cell / Camera objects with stand-in FoV cameras and
    analytic depth maps of the surface (an ellipsoid shell): frustum of the proxy points, proxy-cell registration, signed
    distances to the depth map (grid_sample), view-state / supervision-occupancy / out-of-field updates, surface feature
    reset, occupancy-probability field, coverage gain of five neighbour poses (one with an empty frustum), first strict
    maximum.  The second decision runs on the state the first left behind (view states accumulate, counters grow, stored
    proxy points are refused by Cell.fill).  World points sit on the 2^-6 grid; proxy points are redrawn until no query has
    a k / k+1 neighbour tie and no signed distance sits within 2e-3 of a threshold."""
    import importlib
    from types import SimpleNamespace as NS
    mu = importlib.import_module("macarons.utility.macarons_utils")
    m = _ref_macarons()
    rng = np.random.default_rng(141)
    G = 64.0
    H, W = 64, 114
    x_min, x_max = torch.tensor([-8., -4., -8.]), torch.tensor([8., 4., 8.])
    grid = (2, 1, 2)
    n_proxy = 4000
    axes = np.array([5.5, 2.8, 5.0])
    zfar = 500.
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98, sensor_range=40., min_occ_for_proxy_points=0.1, seq_len=2048,
                use_occ_to_sample_proxy_points=True, jz=False, ddp=False, distance_factor_th=17., image_height=H, image_width=W,
                carving_tolerance=0.05)

    def new_scene(capacity, resolution, feature_dim):
        return mu.Scene(x_min=x_min, x_max=x_max, grid_l=grid[0], grid_w=grid[1], grid_h=grid[2], cell_capacity=capacity,
                        cell_resolution=resolution, n_proxy_points=n_proxy, device="cpu", feature_dim=feature_dim)
    d = rng.standard_normal((2600, 3))
    surf = np.unique(_grid(d / np.linalg.norm(d, axis=1, keepdims=True) * axes + 0.05 * rng.standard_normal((2600, 3)), G), axis=0)
    rng.shuffle(surf)

    def draw_proxy(n):
        q = _grid(rng.uniform(-1, 1, (n, 3)) * [7.9, 3.9, 7.9], G)
        q[q == 0] = 1.0 / G
        return q
    proxy = draw_proxy(n_proxy)
    # the two poses of the trajectory and, per decision, five neighbour poses (the last one looks away from the scene)
    eyes = np.array([[13., 5., -11.], [-12., 6., -9.]], np.float32)
    Rc, Tc = _look_at_target(eyes, np.zeros((2, 3)))
    Pc = np.broadcast_to(_fov_projection(60.0, 1.0, zfar), (2, 4, 4)).copy()
    n_eyes = np.array([[[11., 5., -13.], [14., 3., -8.], [13., 8., -11.], [9., 5., -6.], [30., 0., 0.]],
                       [[-10., 6., -12.], [-14., 4., -6.], [-12., 9., -9.], [-8., 4., -5.], [-30., 0., 0.]]], np.float32)
    n_ats = np.zeros((2, 5, 3), np.float32)
    n_ats[0, 4], n_ats[1, 4] = [90., 0., 0.], [-90., 0., 0.]
    cam = _ref_camera(mu, H, W)
    depth, dmask = [], []
    for c in range(2):
        dd, hh = _ellipsoid_depth(cam.ndc_x_tab.numpy(), cam.ndc_y_tab.numpy(), eyes[c], Rc[c], axes)
        depth.append(dd); dmask.append(hh)
    dts = None

    def tie_scan(ps, surface_scene):
        """Replay compute_scene_occupancy_probability_field's cell loop (and the randperm draws SconeOcc will make) on the
        current state; returns the proxy indices that have a k / k+1 tie in one of their three clouds.  RNG state untouched."""
        state = torch.get_rng_state()
        occ_mask = (ps.proxy_supervision_occ > 0.)[..., 0]
        fov_mask = (ps.out_of_field < 1.)[..., 0]
        cells = ps.get_englobing_cells(ps.proxy_points[occ_mask * fov_mask])
        bad_idx = []
        for cell in cells:
            pcw = surface_scene.get_pt_cloud_from_cells(surface_scene.get_neighboring_cells(cell), return_features=False).numpy()
            _, ind = ps.get_pt_cloud_from_cells(cell, return_features=True)
            cmask = ps.get_proxy_mask_from_indices(ind) * occ_mask
            Xw = ps.proxy_points[cmask].numpy()
            gi = np.nonzero(cmask.numpy())[0]
            if not (pcw.shape[0] > 64 and len(Xw) > 0):
                continue
            M = len(pcw)
            ds = int(np.power(M / (16 * 8), 1. / 2)) or 2
            for lo in range(0, len(Xw), 20000):
                torch.randperm(M)
                p1 = torch.randperm(M).numpy()[:M // ds]
                p2 = torch.randperm(M // ds).numpy()[:(M // ds) // ds]
                pc1 = pcw[p1]; pc2 = pc1[p2]
                Xc = Xw[lo:lo + 20000]
                bad = (_boundary_ties(Xc, pcw, 16, G) | _boundary_ties(Xc, pc1, 16, G) | _boundary_ties(Xc, pc2, 16, G))
                bad_idx += gi[lo:lo + 20000][bad].tolist()
        torch.set_rng_state(state)
        return bad_idx

    real_rand = torch.rand
    for it in range(80):
        torch.manual_seed(5000)
        surface_scene = new_scene(500, 0.2, 1)
        surface_scene.fill_cells(t(surf), features=torch.zeros(len(surf), 1))
        surf_cells = {k: c.cell_pts.numpy().copy() for k, c in surface_scene.cells.items()}
        ps = new_scene(100000, 1e-4, 1)
        ps.initialize_proxy_points()
        ps.proxy_points = t(proxy)
        dts = 3 * ps.distance_between_proxy_points
        out, bad_idx = {}, []
        for c in range(2):
            cam.fov_camera = _StandInCameras(t(Rc[c:c + 1]), t(Tc[c:c + 1]), t(Pc[c:c + 1]), squeeze=True)
            cam.X_cam = t(eyes[c:c + 1])
            cam.fov_camera_0 = cam.fov_camera                                               # testers/scene.py:305
            dmap, dm = t(depth[c]).view(1, H, W, 1), torch.from_numpy(dmask[c]).view(1, H, W, 1)
            torch.manual_seed(5100 + c)
            # ---- testers/scene.py:391-418
            fov_pts, fov_mask = cam.get_points_in_fov(ps.proxy_points, return_mask=True, fov_camera=None, fov_range=params.sensor_range)
            fov_idx = ps.get_proxy_indices_from_mask(fov_mask)
            ps.fill_cells(fov_pts, features=fov_idx.view(-1, 1))
            sgn = cam.get_signed_distance_to_depth_maps(pts=fov_pts, depth_maps=dmap, mask=dm, fov_camera=None)
            ps.update_proxy_view_states(cam, fov_mask, signed_distances=sgn, distance_to_surface=None, X_cam=None)
            ps.update_proxy_supervision_occ(fov_mask, sgn, tol=params.carving_tolerance)
            ps.update_proxy_out_of_field(fov_mask)
            surface_scene.set_all_features_to_value(value=1.)
            sg = sgn.view(-1).numpy()
            near = (np.abs(sg - dts) < 2e-3) | (np.abs(sg + params.carving_tolerance) < 2e-3)
            bad_idx += np.nonzero(fov_mask.numpy())[0][near].tolist()
            bad_idx += tie_scan(ps, surface_scene)
            out[f"fov_mask_{c}"] = np.packbits(fov_mask.numpy())
            out[f"sgn_{c}"] = sg.copy()
            out[f"view_states_{c}"] = np.packbits(ps.view_states.numpy().astype(np.uint8), axis=-1)
            out[f"sup_occ_{c}"] = ps.proxy_supervision_occ.numpy()[:, 0].astype(np.uint8)
            out[f"oof_{c}"] = ps.out_of_field.numpy()[:, 0].astype(np.uint8)
            out[f"n_inside_{c}"] = ps.proxy_n_inside_fov.numpy()[:, 0].astype(np.uint8)
            out[f"n_behind_{c}"] = ps.proxy_n_behind_depth.numpy()[:, 0].astype(np.uint8)
            # ---- :421-425
            with torch.no_grad():
                X_world, vh, occ = mu.compute_scene_occupancy_probability_field(params, m, cam, surface_scene, ps, "cpu")
            out[f"X_world_{c}"], out[f"vh_{c}"], out[f"occ_{c}"] = X_world.numpy(), vh.numpy()[::5].copy(), occ.numpy()
            out[f"proxy_proba_{c}"] = ps.proxy_proba.numpy().copy()
            # ---- :434-454 on five neighbour poses
            Rn, Tn = _look_at_target(n_eyes[c], n_ats[c])
            Pn = np.broadcast_to(_fov_projection(60.0, 1.0, zfar), (5, 4, 4)).copy()
            alln = _StandInCameras(t(Rn), t(Tn), t(Pn))
            out[f"nMview_{c}"], out[f"nMfull_{c}"] = alln.Mv.numpy(), alln.get_full_projection_transform().M.numpy()
            max_gain, next_idx, gains, us = -1., 0, [], []
            for k in range(5):
                fn = _StandInCameras(t(Rn[k:k + 1]), t(Tn[k:k + 1]), t(Pn[k:k + 1]), squeeze=True)
                drawn = []

                def cap(*a, **kw):
                    r_ = real_rand(*a, **kw); drawn.append(r_.numpy().copy()); return r_
                torch.rand = cap
                try:
                    with torch.no_grad():
                        _, _, _, cg = mu.predict_coverage_gain_for_single_camera(
                            params=params, macarons=m, proxy_scene=ps, surface_scene=surface_scene, X_world=X_world,
                            proxy_view_harmonics=vh, occ_probs=occ, camera=cam, X_cam_world=t(n_eyes[c, k:k + 1]), fov_camera=fn)
                finally:
                    torch.rand = real_rand
                gains.append(float(cg.view(-1)[0]))
                us.append(drawn[0].reshape(-1) if drawn else np.zeros(2048, np.float32))
                if cg.shape[0] > 0 and cg > max_gain:
                    max_gain, next_idx = cg, k
            out[f"gains_{c}"], out[f"next_idx_{c}"], out[f"u_{c}"] = np.array(gains, np.float32), np.int64(next_idx), np.stack(us)
            print(f"  decision {c}: {int(fov_mask.sum())} proxy points in fov, field {len(X_world)} points, gains {np.round(gains, 4)}, next {next_idx}")
        bad_idx = sorted(set(bad_idx))
        print(f"  decision golden: pass {it}: {len(bad_idx)} proxy points to redraw")
        if not bad_idx:
            break
        proxy[bad_idx] = draw_proxy(len(bad_idx))
    else:
        raise RuntimeError("no tie-free proxy set found")
    allc = _StandInCameras(t(Rc), t(Tc), t(Pc))
    out.update(x_min=x_min.numpy(), x_max=x_max.numpy(), grid=np.array(grid), surface=surf, proxy=proxy, eyes=eyes, n_eyes=n_eyes,
               Mview=allc.Mv.numpy(), Mfull=allc.get_full_projection_transform().M.numpy(),
               ndc=np.array([cam.min_ndc_x, cam.max_ndc_x, cam.min_ndc_y, cam.max_ndc_y], np.float32), depth=np.stack(depth),
               dmask=np.packbits(np.stack(dmask)), hw=np.array([H, W]), zfar=np.float32(zfar), dts=np.float64(dts),
               box_diag=np.float32(torch.linalg.norm(ps.x_max - ps.x_min).item()))
    for i, (k, v) in enumerate(sorted(surf_cells.items())):
        out[f"cellkey_{i}"] = np.array(eval(k))
        out[f"cellpts_{i}"] = v
    out["n_surface_cells"] = np.int64(len(surf_cells))
    # coverage metrics on the final state (macarons_utils.py:2987-3056): the surface scene as ground truth, a "recovered" scene
    # filled with a noisy subset, and the gain a partial cloud would bring
    rec = new_scene(500, 0.2, 1)
    part = _grid(surf[:900] + 0.08 * rng.standard_normal((900, 3)), G)
    torch.manual_seed(5200)
    rec.fill_cells(t(part), features=torch.zeros(len(part), 1))
    cov, n_gt = surface_scene.scene_coverage(rec, surface_epsilon=0.25)
    surface_scene.set_all_features_to_value(value=0.)
    for k_, c_ in surface_scene.cells.items():                                  # half of the points already covered
        c_.cell_features[::2] = 1.
    part2 = _grid(surf[600:1500] + 0.05 * rng.standard_normal((900, 3)), G)
    cg = surface_scene.camera_coverage_gain(t(part2), surface_epsilon=0.2)
    out.update(cov_part=part, cov_value=np.float64(cov), cov_n=np.int64(n_gt), cov_part2=part2, cov_gain=np.float64(cg), cov_seed=np.int64(5200))
    print(f"  coverage {float(cov):.4f} of {n_gt}; camera coverage gain {float(cg)}")
    save("macarons_decision", **out)



# ---------------------------------------------------------------------------------------------------------------------
# Round 4: a TRAJECTORY (BASELINE config 5: "10 trajectory steps ... achieved surface coverage vs reference").
def _traj_setup():
    """Everything of the trajectory golden that is a function of constants (shared by the generator and, through the .npz,
    the GPU test): pose lattice, neighbour rule, cameras."""
    n_a, n_e = 12, 3
    elev = np.deg2rad([-15.0, 15.0, 40.0])

    def eye_of(a, e):
        a = a % n_a
        th = 2 * np.pi * a / n_a + 0.13
        return np.array([17.0 * np.cos(elev[e]) * np.cos(th), 2.0 + 11.0 * np.sin(elev[e]), 17.0 * np.cos(elev[e]) * np.sin(th)], np.float32)

    def at_of(a, e):
        a = a % n_a
        return np.array([1.5 * np.sin(3 * a + 0.4), 0.4 * (e - 1), 1.5 * np.cos(2 * a + e + 0.2)], np.float32)

    def neighbours(a, e, prev):
        cand = [(a + 1, e), (a - 1, e), (a, e + 1), (a, e - 1), (a + 1, e + 1), (a - 1, e - 1), (a + 2, e)]
        out = []
        for (x, y) in cand:
            x %= n_a
            if 0 <= y < n_e and (x, y) != prev and (x, y) not in out:
                out.append((x, y))
        return out
    return n_a, n_e, eye_of, at_of, neighbours


def gen_trajectory():
    """A TEN-decision trajectory of the MACARONS loop (testers/scene.py:284-454 per pose: ground-truth partial cloud ->
    covered scene -> scene_coverage; partial cloud of the depth map -> surface scene; proxy points in the frustum, carving,
    view states, supervision occupancy; occupancy field; coverage gain of 5-7 neighbour poses; first strict maximum; MOVE to the
    chosen pose) driven on the REAL reference Scene / Cell / Camera objects: 3 x 2 x 3 grid, 24 000 proxy points, surface scene
    empty at the start and growing with every depth map, analytic depth maps of an ellipsoid, stand-in FoV cameras on a pose
    lattice (12 azimuths x 3 elevations; a pose's neighbours = its lattice neighbours minus the pose it came from, plus one pose
    looking away from the scene).  World points sit on the 2^-6 grid (partial clouds are snapped to it -- the harness's sensor
    quantisation -- and stored); hidden draws are keyed by position (keyed_rng.py); proxy points that sit on a numerical decision
    boundary at any step (kNN k / k+1 tie, signed distance within 2e-3 of a threshold, frustum / range boundary, view-state bin
    edge) are redrawn and the trajectory is re-run until none is left."""
    import importlib
    import time as _time
    from types import SimpleNamespace as NS
    import keyed_rng as KR
    from oracle import view_state as V
    mu = importlib.import_module("macarons.utility.macarons_utils")
    m = _ref_macarons()
    rng = np.random.default_rng(151)
    G = 64.0
    H, W = 64, 114
    n_steps = 10
    base_seed = 6100
    x_min, x_max = torch.tensor([-12., -6., -12.]), torch.tensor([12., 6., 12.])
    grid = (3, 2, 3)
    n_proxy = 24000
    axes = np.array([8.0, 3.6, 7.0])
    zfar, gf, eps_cov = 500., 0.9, 0.3
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98, sensor_range=24., min_occ_for_proxy_points=0.1, seq_len=2048,
                use_occ_to_sample_proxy_points=True, jz=False, ddp=False, distance_factor_th=17., image_height=H, image_width=W,
                carving_tolerance=0.05)
    n_a, n_e, eye_of, at_of, neighbours = _traj_setup()
    AWAY = (99, 0)                                   # the extra neighbour of every step: outside the scene, looking away (empty frustum)
    dropped = {}                                     # step -> lattice poses taken off the neighbour list (runner-up within 1e-3 of the best)
    faces = [np.array([-4., 4.]), np.array([0.]), np.array([-4., 4.])]

    def off_faces(q):
        for ax in range(3):                       # Cell.fill's box tests are strict: keep points off the cell faces
            for f in faces[ax]:
                q[q[:, ax] == f, ax] = f + 1.0 / G
        return q

    def new_scene(capacity, resolution, feature_dim, score_threshold=1.):
        return mu.Scene(x_min=x_min, x_max=x_max, grid_l=grid[0], grid_w=grid[1], grid_h=grid[2], cell_capacity=capacity,
                        cell_resolution=resolution, n_proxy_points=n_proxy, device="cpu", feature_dim=feature_dim,
                        score_threshold=score_threshold)
    d = rng.standard_normal((26000, 3))
    gt = off_faces(np.unique(_grid(d / np.linalg.norm(d, axis=1, keepdims=True) * axes, G), axis=0))
    rng.shuffle(gt)

    def draw_proxy(n):
        return off_faces(_grid(rng.uniform(-1, 1, (n, 3)) * [11.9, 5.9, 11.9], G))
    proxy = draw_proxy(n_proxy)
    cam = _ref_camera(mu, H, W)
    ndc = np.array([cam.min_ndc_x, cam.max_ndc_x, cam.min_ndc_y, cam.max_ndc_y], np.float64)
    P1 = _fov_projection(60.0, 1.0, zfar)
    real_rand = torch.rand

    def camera_of(eye, at):
        R, T = _look_at_target(eye[None], at[None])
        return R, T, _StandInCameras(t(R), t(T), t(P1[None]), squeeze=True), _StandInCameras(t(R), t(T), t(P1[None]))

    def frustum_margin(pts, fc, eye):
        """fp64 distance of every point from the decision boundaries of Camera.get_points_in_fov (NDC bounds, z = 0, range)."""
        Mv, Mf = fc.Mv[0].double().numpy(), (fc.Mv[0].double() @ fc.P[0].double()).numpy()
        p4 = np.concatenate((pts.astype(np.float64), np.ones((len(pts), 1))), 1)
        pr, vw = p4 @ Mf, p4 @ Mv
        nx, ny = pr[:, 0] / pr[:, 3], pr[:, 1] / pr[:, 3]
        mg = np.minimum.reduce([np.abs(nx - ndc[0]), np.abs(nx - ndc[1]), np.abs(ny - ndc[2]), np.abs(ny - ndc[3])])
        rg = np.abs(np.linalg.norm(pts.astype(np.float64) - eye.astype(np.float64), axis=1) - params.sensor_range)
        return np.minimum(np.minimum(mg, np.abs(vw[:, 2]) * 1e-1), rg * 1e-1)

    u_fix = {}                                       # (step, cam) -> {sample index: replaced uniform}
    for it in range(60):
        t_pass = _time.time()
        kr = KR.KeyedRandperm(base_seed)
        bad_idx, out = [], {}
        with kr.installed():
            kr.at(-1, "gt_fill")
            gt_scene = new_scene(3000, 0.15, 1)
            gt_scene.fill_cells(t(gt), features=torch.zeros(len(gt), 1))
            covered = new_scene(1500, 0.2, 1)
            surface_scene = new_scene(500, 0.2, 1)
            ps = new_scene(100000, 1e-4, 1, score_threshold=0.95)
            ps.initialize_proxy_points()
            ps.proxy_points = t(proxy)
            dts = 3 * ps.distance_between_proxy_points
            pose, prev = (1, 1), None
            poses, cov_hist = [], []
            for step in range(n_steps):
                eye, at = eye_of(*pose), at_of(*pose)
                Rc, Tc, fc_sq, fc_b = camera_of(eye, at)
                cam.fov_camera, cam.X_cam, cam.fov_camera_0 = fc_sq, t(eye[None]), fc_sq
                dd, hh = _ellipsoid_depth(cam.ndc_x_tab.numpy(), cam.ndc_y_tab.numpy(), eye, Rc[0], axes)
                dmap, dm = t(dd).view(1, H, W, 1), torch.from_numpy(hh).view(1, H, W, 1)
                # ---- testers/scene.py:318-336: ground-truth partial cloud -> covered scene -> coverage
                kr.at(step, "part_gt")
                pg = cam.compute_partial_point_cloud(depth=dmap, mask=dm, fov_cameras=fc_b, gathering_factor=gf, fov_range=params.sensor_range)
                pg_s = off_faces(np.unique(_grid(pg.numpy(), G), axis=0))
                kr.at(step, "covered_fill")
                covered.fill_cells(t(pg_s), features=torch.zeros(len(pg_s), 1))
                cov, n_gt = gt_scene.scene_coverage(covered, surface_epsilon=eps_cov)
                # ---- :371-389: partial cloud of the (perfect) depth map -> surface scene
                kr.at(step, "part")
                pp = cam.compute_partial_point_cloud(depth=dmap, mask=dm, fov_cameras=fc_b, gathering_factor=gf, fov_range=params.sensor_range)
                pp_s = off_faces(np.unique(_grid(pp.numpy(), G), axis=0))
                kr.at(step, "surface_fill")
                surface_scene.fill_cells(t(pp_s), features=torch.zeros(len(pp_s), 1))
                # ---- :391-418
                kr.at(step, "decision")
                fov_pts, fov_mask = cam.get_points_in_fov(ps.proxy_points, return_mask=True, fov_camera=None, fov_range=params.sensor_range)
                fov_idx = ps.get_proxy_indices_from_mask(fov_mask)
                ps.fill_cells(fov_pts, features=fov_idx.view(-1, 1))
                sgn = cam.get_signed_distance_to_depth_maps(pts=fov_pts, depth_maps=dmap, mask=dm, fov_camera=None)
                ps.update_proxy_view_states(cam, fov_mask, signed_distances=sgn, distance_to_surface=None, X_cam=None)
                ps.update_proxy_supervision_occ(fov_mask, sgn, tol=params.carving_tolerance)
                ps.update_proxy_out_of_field(fov_mask)
                surface_scene.set_all_features_to_value(value=1.)
                fm = fov_mask.numpy()
                sg = sgn.view(-1).numpy()
                gi_f = np.nonzero(fm)[0]
                near = (np.abs(sg - dts) < 2e-3) | (np.abs(sg + params.carving_tolerance) < 2e-3)
                bad_idx += gi_f[near].tolist()
                bad_idx += np.nonzero(frustum_margin(proxy, fc_sq, eye) < 2e-5)[0].tolist()
                upd = sg < dts                                               # rays whose view-state bin is written
                if upd.any():
                    mg = V.bin_boundary_margin(proxy[gi_f[upd]][None], eye[None], 7, 14)[0, :, 0]
                    bad_idx += gi_f[upd][mg < 3e-6].tolist()
                k_before = kr.k
                # ---- :421-425
                with torch.no_grad():
                    X_world, vh, occ = mu.compute_scene_occupancy_probability_field(params, m, cam, surface_scene, ps, "cpu")
                lut = {tuple(r_): i_ for i_, r_ in enumerate(proxy.tolist())}
                Xw_np, occ_np = X_world.numpy(), occ.numpy()
                edge = np.nonzero(np.abs(occ_np[:, 0] - params.min_occ_for_proxy_points) < 1e-3)[0]     # `preds > min_occ` is a decision too
                bad_idx += [lut[tuple(r_)] for r_ in Xw_np[edge].tolist()]
                # kNN boundary ties, with the draws SconeOcc actually made (stage 'decision', calls k_before ...)
                occ_mask = (ps.proxy_supervision_occ > 0.)[..., 0]
                seen = (ps.out_of_field < 1.)[..., 0]
                cells = ps.get_englobing_cells(ps.proxy_points[occ_mask * seen])
                draws = [e_ for e_ in kr.log if e_[0] == step and e_[1] == "decision" and e_[2] >= k_before]
                j = 0
                for cell in cells:
                    pcw = surface_scene.get_pt_cloud_from_cells(surface_scene.get_neighboring_cells(cell), return_features=False).numpy()
                    _, ind = ps.get_pt_cloud_from_cells(cell, return_features=True)
                    cmask = ps.get_proxy_mask_from_indices(ind) * occ_mask
                    Xw = ps.proxy_points[cmask].numpy()
                    gi = np.nonzero(cmask.numpy())[0]
                    if not (pcw.shape[0] > 64 and len(Xw) > 0):
                        continue
                    M = len(pcw)
                    ds = int(np.power(M / (16 * 8), 1. / 2)) or 2
                    for lo in range(0, len(Xw), 20000):
                        assert [e_[3] for e_ in draws[3 * j:3 * j + 3]] == [M, M, M // ds], (draws[3 * j:3 * j + 3], M, ds)
                        p1 = kr.peek(step, "decision", draws[3 * j + 1][2], M).numpy()[:M // ds]
                        p2 = kr.peek(step, "decision", draws[3 * j + 2][2], M // ds).numpy()[:(M // ds) // ds]
                        j += 1
                        pc1 = pcw[p1]; pc2 = pc1[p2]
                        Xc = Xw[lo:lo + 20000]
                        bad = (_boundary_ties(Xc, pcw, 16, G) | _boundary_ties(Xc, pc1, 16, G) | _boundary_ties(Xc, pc2, 16, G))
                        bad_idx += gi[lo:lo + 20000][bad].tolist()
                assert 3 * j == len(draws), (j, len(draws))
                # ---- :434-454 on this pose's neighbours (+ one pose looking away: empty frustum)
                nb = [q for q in neighbours(pose[0], pose[1], prev) if q not in dropped.get(step, [])]
                n_eyes = np.stack([eye_of(*q) for q in nb] + [np.array([30., 0., 0.], np.float32)])
                n_ats = np.stack([at_of(*q) for q in nb] + [np.array([90., 0., 0.], np.float32)])
                cam_key = [q[0] * 8 + q[1] for q in nb] + [AWAY[0] * 8]
                Rn, Tn = _look_at_target(n_eyes, n_ats)
                max_gain, next_idx, gains = -1., 0, []
                for k in range(len(n_eyes)):
                    fn = _StandInCameras(t(Rn[k:k + 1]), t(Tn[k:k + 1]), t(P1[None]), squeeze=True)
                    bad_idx_k = np.nonzero(frustum_margin(Xw_np, fn, n_eyes[k]) < 2e-5)[0]
                    bad_idx += [lut[tuple(r_)] for r_ in Xw_np[bad_idx_k].tolist()]      # X_world rows back to proxy indices
                    # Which point a uniform selects hangs on the CDF of the occupancies; the kernels' occupancies differ from the reference's
                    # in the last bits, so a uniform within ~1e-8 of a CDF step would select a neighbouring point there (1 sample of 2048 moves
                    # a gain by up to 5e-4).  Every uniform closer than 1e-6 to a step of the EXACT CDF is moved to the middle of its
                    # interval (same point selected, far from both steps) and recorded; the fp32 sampler of the reference, the fp64 one
                    # of the oracle and the kernels' then agree sample for sample (checked below for the first two).
                    _, km = cam.get_points_in_fov(X_world, return_mask=True, fov_camera=fn, fov_range=params.sensor_range)
                    km = km.numpy()
                    u = KR.keyed_uniforms(base_seed, step, cam_key[k])
                    fix = u_fix.setdefault((step, cam_key[k]), {})
                    for j_, v_ in fix.items():
                        u[j_, 0] = v_
                    kept = km & (occ_np[:, 0] > params.min_occ_for_proxy_points)
                    if kept.any():
                        cn = np.cumsum(occ_np[kept, 0].astype(np.float64))
                        cn /= cn[-1]
                        u64 = u.numpy().reshape(-1).astype(np.float64)
                        ii = np.minimum(np.searchsorted(cn, u64, side="left"), len(cn) - 1)
                        lower, upper = np.where(ii > 0, cn[np.maximum(ii - 1, 0)], 0.0), cn[ii]
                        viol = np.nonzero(np.minimum(upper - u64, u64 - lower) < 1e-6)[0]
                        for j_ in viol.tolist():
                            fix[j_] = float(np.float32((lower[j_] + upper[j_]) / 2))
                            u[j_, 0] = fix[j_]
                    torch.rand = lambda *a, **kw: u.clone()
                    try:
                        with torch.no_grad():
                            pw, _, _, cg = mu.predict_coverage_gain_for_single_camera(
                                params=params, macarons=m, proxy_scene=ps, surface_scene=surface_scene, X_world=X_world,
                                proxy_view_harmonics=vh, occ_probs=occ, camera=cam, X_cam_world=t(n_eyes[k:k + 1]), fov_camera=fn)
                    finally:
                        torch.rand = real_rand
                    if km.any():
                        res, _, inv, _ = V.sample_proxy_points(Xw_np[km], occ_np[km], np.zeros((int(km.sum()), 1), np.float32), u.numpy().reshape(-1),
                                                               params.min_occ_for_proxy_points, exact=True)
                        assert np.array_equal(res[inv], pw[0].numpy()), "the reference's sampler and the exact CDF disagree despite the margin"
                    gains.append(float(cg.view(-1)[0]))
                    if cg.shape[0] > 0 and cg > max_gain:
                        max_gain, next_idx = cg, k
                order_g = np.argsort(np.array(gains))[::-1]
                if gains[order_g[0]] - gains[order_g[1]] <= 1e-3 * gains[order_g[0]]:
                    # two candidates within 1e-3: the decision would hang on the last digits; the harness takes the runner-up off the
                    # list (as a collision test would) and the trajectory is re-run
                    dropped.setdefault(step, []).append(nb[order_g[1]])
                    print(f"    step {step}: runner-up {nb[order_g[1]]} within 1e-3 of the best: dropped from the neighbour list, re-run")
                    bad_idx.append(-1)
                    break
                assert next_idx < len(nb)
                out[f"nb_{step}"] = np.array(nb + [AWAY], np.int32)
                alln = _StandInCameras(t(Rn), t(Tn), t(np.broadcast_to(P1, (len(Rn), 4, 4)).copy()))
                out[f"Mview_{step}"], out[f"Mfull_{step}"] = fc_b.Mv.numpy()[0], fc_b.get_full_projection_transform().M.numpy()[0]
                out[f"eye_{step}"], out[f"n_eyes_{step}"] = eye, n_eyes
                out[f"nMview_{step}"], out[f"nMfull_{step}"] = alln.Mv.numpy(), alln.get_full_projection_transform().M.numpy()
                out[f"depth_{step}"], out[f"dmask_{step}"] = dd, np.packbits(hh)
                out[f"fov_mask_{step}"] = np.packbits(fm)
                out[f"sup_occ_{step}"] = np.packbits(ps.proxy_supervision_occ.numpy()[:, 0].astype(np.uint8))
                out[f"oof_{step}"] = np.packbits(ps.out_of_field.numpy()[:, 0].astype(np.uint8))
                out[f"vs_rowsum_{step}"] = ps.view_states.numpy().sum(-1).astype(np.uint8)
                out[f"n_inside_{step}"] = ps.proxy_n_inside_fov.numpy()[:, 0].astype(np.uint8)
                out[f"n_behind_{step}"] = ps.proxy_n_behind_depth.numpy()[:, 0].astype(np.uint8)
                out[f"field_n_{step}"] = np.int64(len(X_world))
                out[f"occ_{step}"] = occ_np[::7, 0].copy()
                out[f"vh_{step}"] = vh.numpy()[::37].copy()
                out[f"gains_{step}"] = np.array(gains, np.float32)
                out[f"part_gt_{step}"] = np.round(pg_s * G).astype(np.int16)
                out[f"part_{step}"] = np.round(pp_s * G).astype(np.int16)
                out[f"part_raw_n_{step}"] = np.array([len(pg), len(pp)], np.int64)
                out[f"surface_n_{step}"] = np.array([len(c.cell_pts) for _, c in sorted(surface_scene.cells.items())], np.int32)
                cov_hist.append(float(cov))
                poses.append((pose, next_idx, nb))
                print(f"  step {step}: pose {pose} cov {float(cov):.4f} fov {int(fm.sum())} field {len(X_world)} surface {sum(len(c.cell_pts) for c in surface_scene.cells.values())} "
                      f"gains {np.round(gains, 3)} -> {next_idx} {nb[next_idx]}")
                prev, pose = pose, nb[next_idx]
        rerun = -1 in bad_idx
        bad_idx = sorted(set(bad_idx) - {-1})
        print(f"  trajectory golden: pass {it}: {len(bad_idx)} proxy points to redraw ({_time.time() - t_pass:.0f} s)")
        if not bad_idx and not rerun:
            break
        proxy[bad_idx] = draw_proxy(len(bad_idx))
    else:
        raise RuntimeError("no boundary-free proxy set found")
    sizes = kr.sizes()
    stages = ["part_gt", "covered_fill", "part", "surface_fill", "decision"]
    flat, off = KR.pack_sizes({k_: v_ for k_, v_ in sizes.items() if k_[0] >= 0}, stages)
    fix_rows = np.array([[s_, k_, j_] for (s_, k_), d_ in sorted(u_fix.items()) for j_ in sorted(d_)], np.int32).reshape(-1, 3)
    fix_vals = np.array([u_fix[(s_, k_)][j_] for s_, k_, j_ in fix_rows.tolist()], np.float32)
    out.update(x_min=x_min.numpy(), x_max=x_max.numpy(), grid=np.array(grid), gt=np.round(gt * G).astype(np.int16), gt_fill_sizes=np.array(sizes[(-1, "gt_fill")], np.int32),
               proxy=np.round(proxy * G).astype(np.int16), G=np.float32(G), axes=axes, hw=np.array([H, W]), zfar=np.float32(zfar), gf=np.float32(gf),
               eps_cov=np.float32(eps_cov), P=P1, sensor_range=np.float32(params.sensor_range), base_seed=np.int64(base_seed), n_steps=np.int64(n_steps), ndc=ndc.astype(np.float32),
               ndc_x_tab=cam.ndc_x_tab.numpy(), ndc_y_tab=cam.ndc_y_tab.numpy(), dts=np.float64(dts),
               rng_sizes=flat, rng_off=off, u_fix_rows=fix_rows, u_fix_vals=fix_vals,
               coverage=np.array(cov_hist, np.float64), cov_n=np.int64(n_gt), pose_a=np.array([p_[0][0] for p_ in poses]), pose_e=np.array([p_[0][1] for p_ in poses]),
               next_idx=np.array([p_[1] for p_ in poses], np.int64), view_states_final=np.packbits(ps.view_states.numpy().astype(np.uint8), axis=-1),
               proxy_proba_final=ps.proxy_proba.numpy()[:, 0].copy())
    save("macarons_trajectory", **out)


GROUPS = {"masked": gen_masked, "trajectory": gen_trajectory, "decision": gen_decision, "occ_field": gen_occ_field, "formats": gen_formats, "e2e_grid": gen_e2e_grid, "fov": gen_fov, "distance": gen_distance, "wrapper": gen_macarons_wrapper, "single_camera": gen_single_camera, "cell": gen_cell, "unproject": gen_unproject, "viewspace": gen_viewspace, "filter": gen_filter, "macarons": gen_macarons, "e2e": gen_e2e, "view": gen_view, "scorer": gen_scorer, "sh": gen_sh, "knn": gen_knn, "blocks": gen_blocks, "vis": gen_vis, "occ": gen_occ}

if __name__ == "__main__":
    todo = sys.argv[1:] or list(GROUPS)
    for g in todo:
        print("==", g)
        GROUPS[g]()
