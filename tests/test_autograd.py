"""The composite-torch functions behind the HIP forward passes' gradients (macarons_amd/autograd.py), on CPU:
their values against the reference's goldens (so they ARE the reference's functions) and their gradients against fp64
finite differences."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import weights  # noqa: E402
from macarons_amd import autograd as A  # noqa: E402


def _model(cls, seed, shift=0.0):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls()
    sd = weights.make_state_dict(weights.shapes_of(m), seed)
    if shift:
        sd["linear3.bias"] = sd["linear3.bias"] + np.float32(shift)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval()


def t(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def test_sh_basis_matches_reference_convention():
    g = golden("sh_basis")
    th, ph = t(g["theta"], torch.float64), t(g["phi"], torch.float64)
    n = torch.stack((torch.sin(th) * torch.sin(ph), torch.cos(th), torch.sin(th) * torch.cos(ph)), -1)     # Y-up, azimuth from +Z toward +X
    assert np.abs(A.sh_basis(n).numpy() - g["Y_f64"]).max() < 1e-12


@pytest.mark.parametrize("name", ["scorer_b1_n2048_c20", "scorer_b2_n500_c7"])
def test_scorer_composite_matches_reference(name):
    d = golden(name)
    for sfx, sig in (("sig", True), ("relu", False)):
        a = (t(d["pts"], torch.float64), t(d["harmonics"], torch.float64), t(d["cams"], torch.float64), sig)
        assert rel_err(A.coverage_gain(*a).numpy(), d["gain64_" + sfx]) < 1e-12
        assert np.abs(A.visibilities(*a).numpy() - d["vis64_" + sfx]).max() < 1e-9
        a32 = (t(d["pts"]), t(d["harmonics"]), t(d["cams"]), sig)
        assert rel_err(A.coverage_gain(*a32).numpy(), d["gain32_" + sfx]) < 1e-4


def test_network_composites_match_reference():
    from macarons_amd.networks import SconeVis, SconeOcc
    vis, occ = _model(SconeVis, 1), _model(SconeOcc, 2)
    g = golden("scone_vis")
    with torch.no_grad():
        for N in (16, 333):
            assert rel_err(A.scone_vis(vis, t(g[f"pts_{N}"]), t(g[f"vh_{N}"])).numpy(), g[f"y_{N}"]) < 1e-4
        y = A.scone_vis(vis, t(g["pts_b3"]), t(g["vh_b3"]))
        assert rel_err(y.numpy(), g["y_b3"]) < 1e-4
        # padded batch == sliced clouds
        lens = torch.tensor([100, 37, 5])
        yp = A.scone_vis(vis, t(g["pts_b3"]), t(g["vh_b3"]), lens)
        for b, n in enumerate(lens.tolist()):
            ys = A.scone_vis(vis, t(g["pts_b3"][b:b + 1, :n]), t(g["vh_b3"][b:b + 1, :n]))
            assert torch.allclose(yp[b, :n], ys[0], atol=2e-6)
    g = golden("scone_occ")
    from oracle import knn
    for tag in ("m100_q17", "m1024_q300"):
        pc, x, vh = g[f"{tag}_pc"], g[f"{tag}_x"], g[f"{tag}_vh"]
        perms = [g[f"{tag}_perm{i}"].astype(np.int64) for i in range(3)]
        scales = [pc, pc[:, perms[1]]]
        scales.append(scales[1][:, perms[2]])
        idx = [torch.from_numpy(knn.knn_points(x, s_, 16)[2]) for s_ in scales]
        with torch.no_grad():
            y = A.scone_occ(occ, t(pc[:, perms[0]]), [t(s_) for s_ in scales], t(x), t(vh), idx)
        assert rel_err(y.numpy(), g[f"{tag}_y"]) < 1e-4


def _fd_check(fn, x, eps=1e-6, n_probe=12, seed=0):
    """max |analytic - central difference| over random probes of the scalar function fn(x) (fp64)."""
    x = x.clone().requires_grad_(True)
    y = fn(x)
    (gx,) = torch.autograd.grad(y, x)
    rng = np.random.default_rng(seed)
    flat = x.detach().reshape(-1)
    worst = 0.0
    for k in rng.choice(flat.numel(), min(n_probe, flat.numel()), replace=False):
        xp, xm = flat.clone(), flat.clone()
        xp[k] += eps; xm[k] -= eps
        fd = (fn(xp.view_as(x)) - fn(xm.view_as(x))) / (2 * eps)
        worst = max(worst, abs(float(fd) - float(gx.reshape(-1)[k])) / max(1e-6, abs(float(fd))))
    return worst


def test_gradients_match_finite_differences():
    from macarons_amd.networks import SconeVis, SconeOcc
    rng = np.random.default_rng(3)
    vis, occ = _model(SconeVis, 1).double(), _model(SconeOcc, 2).double()
    w = t(rng.standard_normal((1, 5)), torch.float64)
    pts = t(np.concatenate([rng.uniform(-.5, .5, (1, 9, 3)), rng.uniform(.1, 1, (1, 9, 1))], -1), torch.float64)
    harm = t(rng.standard_normal((1, 9, 64)) * 0.5, torch.float64)
    cams = t(rng.standard_normal((1, 5, 3)), torch.float64)
    vh = t(rng.standard_normal((1, 9, 64)) * 0.3, torch.float64)
    assert _fd_check(lambda h: (A.coverage_gain(pts, h, cams) * w).sum(), harm) < 1e-6
    assert _fd_check(lambda p: (A.coverage_gain(p, harm, cams) * w).sum(), pts) < 1e-6
    assert _fd_check(lambda p: (A.coverage_gain(pts, A.scone_vis(vis, p, vh), cams) * w).sum(), pts) < 1e-5      # the trainers' chain
    # a parameter of SconeVis
    p0 = vis.fc3.weight
    assert _fd_check(lambda W: (A.coverage_gain(pts, torch.nn.functional.linear(
        torch.nn.functional.gelu(torch.randn(1, 9, 128, dtype=torch.float64, generator=torch.Generator().manual_seed(1))), W, vis.fc3.bias),
        cams) * w).sum(), p0.detach()) < 1e-6
    # SconeOcc: gradient w.r.t. the query points through the neighbour offsets and the x-embedding
    pc = t(rng.uniform(-.3, .3, (1, 40, 3)), torch.float64)
    x = t(rng.uniform(-.3, .3, (1, 4, 3)), torch.float64)
    vq = t(rng.standard_normal((1, 4, 64)) * 0.3, torch.float64)
    scales = [pc, pc[:, :20], pc[:, :18]]
    idx = [torch.cdist(x, s_).topk(16, largest=False)[1] for s_ in scales]
    assert _fd_check(lambda q: A.scone_occ(occ, pc[:, :32], scales, q, vq, idx).sum(), x, eps=1e-7) < 1e-4
