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


GROUPS = {"scorer": gen_scorer, "sh": gen_sh, "knn": gen_knn}

if __name__ == "__main__":
    todo = sys.argv[1:] or list(GROUPS)
    for g in todo:
        print("==", g)
        GROUPS[g]()
