"""Parity of the HIP SH coverage-gain scorer (through the C ABI) against the oracle and the
reference goldens.  Tolerances (north_star): gains within 1e-4 relative (fp32)."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_err
from oracle import scorer, cport

pytestmark = pytest.mark.gpu


def _run(dev, pts, harm, cams, use_sigmoid=True, **kw):
    from macarons_amd import ops
    g = ops.sh_coverage_gain(torch.from_numpy(pts).to(dev), torch.from_numpy(harm).to(dev),
                             torch.from_numpy(cams).to(dev), use_sigmoid, **kw)
    v = ops.sh_visibilities(torch.from_numpy(pts).to(dev), torch.from_numpy(harm).to(dev),
                            torch.from_numpy(cams).to(dev), use_sigmoid)
    torch.cuda.synchronize()
    return g.cpu().numpy(), v.cpu().numpy()


@pytest.mark.parametrize("name", ["scorer_b1_n2048_c20", "scorer_b2_n500_c7"])
@pytest.mark.parametrize("sfx,use_sigmoid", [("sig", True), ("relu", False)])
def test_golden(dev, name, sfx, use_sigmoid):
    d = golden(name)
    g, v = _run(dev, d["pts"], d["harmonics"], d["cams"], use_sigmoid)
    assert g.shape == d["gain32_" + sfx].shape and v.shape == d["vis32_" + sfx].shape
    assert rel_err(g, d["gain32_" + sfx]) < 1e-4          # the north_star bar, vs the fp32 reference
    assert rel_err(g, d["gain64_" + sfx]) < 2e-5          # and much closer to the fp64 reference
    # per-point values: vs the reference's fp64 run (its fp32 trig is ill-conditioned)
    scale = max(1.0, np.abs(d["vis64_" + sfx]).max())
    assert np.abs(v - d["vis64_" + sfx]).max() < 2e-5 * scale


@pytest.mark.parametrize("B,N,C,P", [(1, 1, 1, 3), (1, 63, 3, 3), (3, 257, 65, 4), (2, 1000, 130, 5), (1, 5000, 52, 4)])
def test_ragged_shapes_vs_oracle(dev, B, N, C, P):
    rng = np.random.default_rng(B * 1000 + N + C)
    pts = rng.uniform(-.5, .5, (B, N, P)).astype(np.float32)
    harm = (rng.standard_normal((B, N, 64)) * 0.7).astype(np.float32)
    cams = rng.standard_normal((B, C, 3)).astype(np.float32)
    cams = (1.5 * cams / np.linalg.norm(cams, axis=-1, keepdims=True)).astype(np.float32)
    g, v = _run(dev, pts, harm, cams)
    ref_v = scorer.compute_visibilities(pts, harm, cams, True, "trigfree", np.float64)
    assert np.abs(v - ref_v).max() < 2e-5
    assert rel_err(g, ref_v.mean(-1)) < 2e-5
    # C port (the literal fp32 restatement, also the timed CPU baseline)
    gp, _ = cport.coverage_gain(pts, harm, cams)
    assert rel_err(g, gp) < 1e-4


def test_partition_invariance_and_determinism(dev):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-.5, .5, (1, 3000, 4)).astype(np.float32)
    harm = rng.standard_normal((1, 3000, 64)).astype(np.float32)
    cams = rng.standard_normal((1, 77, 3)).astype(np.float32)
    g0, _ = _run(dev, pts, harm, cams)
    for wps in (1, 2, 3, 8):
        g, _ = _run(dev, pts, harm, cams, waves_per_simd=wps)
        assert np.array_equal(g, g0)        # per-camera results do not depend on the work partition
    g1, _ = _run(dev, pts, harm, cams)
    assert np.array_equal(g0, g1)           # bit-stable run to run (no float atomics)


def test_known_answers(dev):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-.5, .5, (2, 777, 4)).astype(np.float32)
    cams = rng.standard_normal((2, 9, 3)).astype(np.float32)
    g, v = _run(dev, pts, np.zeros((2, 777, 64), np.float32), cams)
    assert np.all(g == 0.5) and np.all(v == 0.5)           # sigmoid(0) for every camera
    # permutation invariance over points
    harm = rng.standard_normal((2, 777, 64)).astype(np.float32)
    g0, _ = _run(dev, pts, harm, cams)
    perm = rng.permutation(777)
    g1, _ = _run(dev, pts[:, perm].copy(), harm[:, perm].copy(), cams)
    assert rel_err(g1, g0) < 1e-6


def test_axis_aligned_rays_are_finite(dev):
    # ray exactly on +-Y (x = z = 0): the build returns the finite limit (all m != 0 terms vanish)
    pts = np.zeros((1, 4, 3), np.float32)
    cams = np.array([[[0, 1.5, 0], [0, -1.5, 0], [1.5, 0, 0], [0, 0, -1.5]]], np.float32)
    harm = np.random.default_rng(0).standard_normal((1, 4, 64)).astype(np.float32)
    g, v = _run(dev, pts, harm, cams)
    assert np.isfinite(g).all() and np.isfinite(v).all()
    ref = scorer.compute_visibilities(pts, harm, cams, True, "trigfree", np.float64)
    assert np.abs(v - ref).max() < 2e-5


@pytest.mark.parametrize("B,N,C", [(1, 100_000, 200), (1, 16_384, 100), (8, 32_768, 200), (1, 100_000, 512)])
def test_large_headline_properties(dev, B, N, C):
    """BASELINE.json sizes (headline N=100k / C=200; configs 2, 3, 4): size-independent properties instead of a full oracle run."""
    from macarons_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(1234)
    pts = (torch.rand(B, N, 4, generator=gen) - 0.5)
    harm = torch.randn(B, N, 64, generator=gen) * 0.5
    cams = torch.randn(B, C, 3, generator=gen)
    cams = 1.5 * cams / cams.norm(dim=-1, keepdim=True)
    g = ops.sh_coverage_gain(pts.to(dev), harm.to(dev), cams.to(dev))
    # mean of per-point visibilities == gain
    v = ops.sh_visibilities(pts.to(dev), harm.to(dev), cams.to(dev))
    assert rel_err(g.cpu().numpy(), v.double().mean(-1).cpu().numpy()) < 1e-6
    # splitting the cloud in two halves: gain = average of the halves' gains
    h = N // 2
    ga = ops.sh_coverage_gain(pts[:, :h].contiguous().to(dev), harm[:, :h].contiguous().to(dev), cams.to(dev))
    gb = ops.sh_coverage_gain(pts[:, h:].contiguous().to(dev), harm[:, h:].contiguous().to(dev), cams.to(dev))
    assert rel_err(((ga + gb) / 2).cpu().numpy(), g.cpu().numpy()) < 1e-6
    # a bounded sample against the C port (first cloud, 7 cameras)
    sel = slice(0, 7)
    gp, _ = cport.coverage_gain(pts[:1].numpy(), harm[:1].numpy(), cams[:1, sel].numpy())
    assert rel_err(g[:1, sel].cpu().numpy(), gp) < 1e-4
    assert 0.0 < float(g.min()) and float(g.max()) < 1.0
    # the decision record == torch.max over the cameras
    rec = ops.best_record(g)
    ref = torch.max(g, dim=1)
    assert torch.equal(rec[:, 0], ref.values) and torch.equal(rec[:, 1].long(), ref.indices)


def test_random_shapes_and_tile_edges(dev):
    """48 random shapes (tools/fuzz_scorer.py runs 600): clouds of 1 .. 5000 points with every wave-tile edge, 1 .. 300 cameras,
    1 .. 3 clouds, point stride 3 / 4, both activations, gains and per-point visibilities -- within 2e-5 of the C port of the
    reference scorer, or, where two fp32 evaluations of a cancelling sum are further apart than that (small clouds, relu), within
    2e-6 of the fp64 evaluation."""
    rng = np.random.default_rng(77)
    edge_n = [1, 2, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 4095, 4096, 4097]
    for case in range(48):
        B = int(rng.integers(1, 4))
        N = edge_n[(case // 2) % len(edge_n)] if case % 2 == 0 else int(rng.integers(1, 5000))
        C = int(rng.choice([1, 2, 7, 20, 52, 64, 100, 200, 300]))
        P = int(rng.choice([3, 4]))
        sig = bool(rng.integers(0, 2))
        pts = rng.uniform(-.5, .5, (B, N, P)).astype(np.float32)
        harm = (rng.standard_normal((B, N, 64)) * rng.choice([0.1, 0.5, 1.5])).astype(np.float32)
        cams = rng.standard_normal((B, C, 3)).astype(np.float32)
        cams = (1.5 * cams / np.linalg.norm(cams, axis=-1, keepdims=True)).astype(np.float32)
        ref, _ = cport.coverage_gain(pts, harm, cams, use_sigmoid=sig)
        g, v = _run(dev, pts, harm, cams, sig)
        assert np.isfinite(g).all() and v.shape == (B, C, N), (B, N, C, P, sig)
        scale = max(1e-6, float(np.abs(ref).max()))
        e_g, e_v = np.abs(g - ref).max() / scale, np.abs(v.mean(-1) - ref).max() / scale
        if max(e_g, e_v) > 2e-5:
            truth = scorer.compute_coverage_gain(pts[..., :3], harm, cams, use_sigmoid=sig, dtype=np.float64)
            scale = max(1e-6, float(np.abs(truth).max()))
            assert np.abs(g - truth).max() / scale < 2e-6 and np.abs(v.mean(-1) - truth).max() / scale < 2e-6, (B, N, C, P, sig)


def test_empty_dimensions_follow_upstream(dev):
    """No points / no cameras / no clouds: upstream's tensor algebra returns NaN gains for an empty cloud (sum over nothing / 0,
    SconeVis.py:250) and empty tensors otherwise (checked against the imported reference when this test was written: shapes
    (1,3) NaN, (2,0), (0,3); visibilities (B,C,N); the n-tuple form (B, C^n) + its index table)."""
    from macarons_amd.networks import SconeVis
    m = SconeVis().to(dev).eval()
    for (B, N, C) in [(1, 0, 3), (2, 5, 0), (0, 5, 3)]:
        pts, h, cams = torch.rand(B, N, 4, device=dev), torch.rand(B, N, 64, device=dev), torch.rand(B, C, 3, device=dev)
        with torch.no_grad():
            g = m.compute_coverage_gain(pts, h, cams)
            v = m.compute_visibilities(pts, h, cams)
            g2, idx2 = m.compute_coverage_gain_multiple(pts, h, cams, 2)
        assert tuple(g.shape) == (B, C) and tuple(v.shape) == (B, C, N) and tuple(g2.shape) == (B, C * C) and tuple(idx2.shape) == (C * C, 2)
        assert g.device.type == "cuda" and (g.numel() == 0 or bool(torch.isnan(g).all())) and (g2.numel() == 0 or bool(torch.isnan(g2).all()))


def test_gain_best_is_the_gains_plus_torch_max(dev):
    """mcr_sh_coverage_gain_best (one call: gains + the arg-max record of testers/shapenet.py:172) == mcr_sh_coverage_gain + torch.max,
    through the ctypes wrapper and through torch.ops.macarons; a NaN gain wins the arg-max like torch.max."""
    from macarons_amd import ops
    import macarons_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(5)
    for B, N, C, sig in ((1, 2048, 20, True), (2, 500, 7, True), (1, 100_000, 200, True), (3, 65, 300, False), (1, 1, 1, True)):
        pts = torch.from_numpy(rng.uniform(-.5, .5, (B, N, 4)).astype(np.float32)).to(dev)
        harm = torch.from_numpy((rng.standard_normal((B, N, 64)) * 0.5).astype(np.float32)).to(dev)
        cams = rng.standard_normal((B, C, 3)).astype(np.float32)
        cams = torch.from_numpy((1.5 * cams / np.linalg.norm(cams, axis=-1, keepdims=True)).astype(np.float32)).to(dev)
        want = ops.sh_coverage_gain(pts, harm, cams, sig)
        g1, r1 = ops.sh_coverage_gain_best(pts, harm, cams, sig)
        g2, r2 = torch.ops.macarons.sh_coverage_gain_best(pts, harm, cams, sig)
        assert torch.equal(g1, want) and torch.equal(g2, want) and torch.equal(r1, r2), (B, N, C)
        ref = torch.max(want, dim=1)
        assert torch.equal(r1[:, 0], ref.values) and torch.equal(r1[:, 1], ref.indices.float()), (B, N, C)
    bad = harm.clone()
    bad[0, 0, 5] = float("nan")
    gb, rb = ops.sh_coverage_gain_best(pts, bad, cams, True)
    ref = torch.max(gb, dim=1)
    assert bool(torch.isnan(rb[0, 0])) and float(rb[0, 1]) == float(ref.indices[0])


def test_steps_in_flight_on_two_streams_are_independent(dev):
    """bench.py issues consecutive scorer steps round-robin on two streams (each call allocates its outputs and scratch from the
    stream-aware allocator, nothing is shared between calls): 40 steps over 4 different camera sets, two in flight at any time, return
    bit for bit what the same calls return one after the other on one stream."""
    import macarons_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(9)
    N, C = 100_000, 200
    pts = torch.from_numpy(rng.uniform(-.5, .5, (1, N, 4)).astype(np.float32)).to(dev)
    harm = torch.from_numpy((rng.standard_normal((1, N, 64)) * 0.5).astype(np.float32)).to(dev)
    cam_sets = []
    for _ in range(4):
        c = rng.standard_normal((1, C, 3)).astype(np.float32)
        cam_sets.append(torch.from_numpy((1.5 * c / np.linalg.norm(c, axis=-1, keepdims=True)).astype(np.float32)).to(dev))
    want = [torch.ops.macarons.sh_coverage_gain_best(pts, harm, c, True) for c in cam_sets]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    got = []
    for i in range(40):
        with torch.cuda.stream(streams[i % 2]):
            got.append(torch.ops.macarons.sh_coverage_gain_best(pts, harm, cam_sets[i % 4], True))
    torch.cuda.synchronize()
    for i, (g, r) in enumerate(got):
        assert torch.equal(g, want[i % 4][0]) and torch.equal(r, want[i % 4][1]), i
