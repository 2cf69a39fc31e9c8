"""Variant 7 -- BASELINE.json config 3 as it is named ("bf16": a 16-bit matrix path) -- the OPT-IN numerics of the network path:
ONE fp16 plane per matrix operand (one MFMA per product, fp32 accumulation) in the fused local transformers (local_pct7.hip), the
SconeOcc head (linear3p.hip, single-plane forms) and -- for sequences of >= 512 tokens -- the encoders of SconeVis and of SconeOcc's
global transformer (their GEMMs and attention_planes.hip's single-plane form); LayerNorm statistics, attention scores and soft-max,
GELU, pooling, the SH scorer and every reduction stay fp32.  Selected per call only (`with ops.variant(7)` / mcr_call_variant(7)); never a process default.

ITS OWN TOLERANCE, measured here and stated -- NOT the 1e-4 of variants 1 / 5 / 6.  Occupancies (relative, max-norm, against the fp64
oracle on the inputs of scone_occ.npz; measured value -> asserted bound, per weight set):
    golden weights (seed 2)   1.0e-3 -> 2e-3   (also against the REFERENCE's own fp32 output: the same)
    seed 11                   8.1e-4 -> 2e-3
    seed 7                    4.7e-3 -> 8e-3   (this network amplifies operand errors 8x more: variant 6 has 7e-6 there, 9e-7-1.4e-6 on the others)
    seed 2, local weights x4  3.5e-3 -> 6e-3   (pooled local features alone: 9e-2, variant 6: 8e-5 -- soft-max logits 16x larger)
SconeVis harmonics (2048 points, fp64 oracle): measured -> VIS_TOL = 3e-3.
The bound is a property of (path, checkpoint): a network amplifies ANY operand error by its own factor (round 5's pricing table).  What
is checkpoint-INDEPENDENT is the ratio: variant 7 carries 11 of variant 6's 22 operand bits, so its error stays below AMPLIFICATION =
2^12 times variant 6's on the same inputs (measured ratios 340 .. 2100); asserted for every weight set.

DECISIONS.  What must hold exactly is the arg-max camera on every golden decision of the reference -- the two grid decisions (configs 1,
2) and the ten decisions of the trajectory golden (config 5) -- and it does (margins between the best and second camera there: 2.4 % ..
19 % of the largest gain).  What does not survive a 1e-3 change of the occupancies untouched is the identity of the sampled proxy set:
inverse-CDF sampling is a step function of the cumulative occupancies, so a few of the 2048 uniforms land on a neighbouring point --
measured 48-179 of 2048 samples (2.3-8.7 %), 91-99 % of the unique points in common with the reference's set (asserted: <= 15 % /
>= 85 %; the counts move with every change of the kernel's rounding pattern: they are a property of the CDF's steps, not of the kernel).
The gains are a Monte-Carlo estimate over that sample and move with it: measured 0.4-3.3e-2 relative end to end (bound asserted:
GAIN_E2E_TOL = 5e-2) while the same networks on the REFERENCE's sampled set reproduce its gains at 5e-4 .. 7e-4 (GAIN_TOL = 2e-3).  All of this is REPORTED
(gpurun_out/variant7_report.json, printed with -s), not hidden behind a looser comparison."""
import contextlib
import io
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, rel_err, ROOT

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import weights  # noqa: E402
from oracle import nets  # noqa: E402

pytestmark = pytest.mark.gpu
OCC_TOL = 2e-3        # |occ - occ_ref| / max |occ_ref| on the golden weights (the table above for other weight sets)
OCC_TOL_BY_WEIGHTS = {(2, 1.0): 2e-3, (11, 1.0): 2e-3, (7, 1.0): 8e-3, (2, 4.0): 6e-3}
VIS_TOL = 3e-3        # SconeVis harmonics vs the fp64 oracle (encoders + attention on one plane)
GAIN_E2E_TOL = 5e-2   # gains of a whole decision: Monte-Carlo noise of a different sampled set (measured <= 3.3e-2)
GAIN_TOL = 2e-3       # gains on the SAME sampled set
LOCAL_TOL = 2e-3      # pooled local features of one fused transformer vs the fp64 oracle (measured: 6e-4 .. 1.3e-3 at unit scale)
AMPLIFICATION = 4096.0   # variant 7's error <= 2^12 x variant 6's on the same inputs and weights (measured ratios: 340 .. 2100)
REPORT = {}


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _report(key, value):
    REPORT[key] = value
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join(ROOT, "gpurun_out", "variant7_report.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old[key] = value
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print(f"[variant 7] {key}: {json.dumps(value)}")


def _occ(dev, seed, local_scale=1.0):
    from macarons_amd.networks import SconeOcc
    with contextlib.redirect_stdout(io.StringIO()):
        m = SconeOcc()
    sd = weights.make_state_dict(weights.shapes_of(m), seed)
    if local_scale != 1.0:          # trained networks are not unit-scale: the reference's init gives |activation| ~ 4 in these layers
        sd = {k: (v * np.float32(local_scale) if (k.startswith("local_transformers.") and k.endswith("weight") and v.ndim == 2) else v)
              for k, v in sd.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def test_variant_7_is_per_call_only(dev):
    """Never a process default: mcr_set_local_pct_variant(7) is refused, the default stays what it was; ops.variant(7) scopes it."""
    import ctypes
    from macarons_amd import ops, _lib
    L = _lib.lib()
    before = L.mcr_get_local_pct_variant()
    assert L.mcr_set_local_pct_variant(ctypes.c_int(7)) != 0 and L.mcr_get_local_pct_variant() == before
    with ops.variant(7):
        assert ops.current_variant() == 7
    assert ops.current_variant() == before


@pytest.mark.parametrize("seed,local_scale", [(2, 1.0), (7, 1.0), (11, 1.0), (2, 4.0)])
def test_fused_local_transformer_single_plane(dev, seed, local_scale):
    """local_pct7_kernel vs the fp64 oracle, ragged workgroup counts, three weight seeds + the 4x local-weights case."""
    from macarons_amd import ops
    from macarons_amd.networks.packing import pack_local_pct
    m, sd = _occ(dev, seed, local_scale)
    rng = np.random.default_rng(4 + seed)
    worst, worst6 = 0.0, 0.0
    for S in (1, 3, 4, 1001):
        offs = (rng.standard_normal((S, 16, 3)) * (0.05 if S != 3 else 0.5)).astype(np.float32)
        for sc in range(3):
            with torch.no_grad():
                with ops.variant(7):
                    fused = ops.local_pct_forward(T(offs, dev), pack_local_pct(m.local_transformers[sc], 7)).cpu().numpy()
                with ops.variant(6):
                    fused6 = ops.local_pct_forward(T(offs, dev), pack_local_pct(m.local_transformers[sc], 6)).cpu().numpy()
            ref = nets.pc_transformer(sd, f"local_transformers.{sc}.", offs, np.float64)
            assert np.isfinite(fused).all()
            worst, worst6 = max(worst, rel_err(fused, ref)), max(worst6, rel_err(fused6, ref))
    _report(f"local_transformer_rel_err_seed{seed}_x{local_scale:g}", {"variant7": worst, "variant6": worst6})
    assert worst < AMPLIFICATION * max(worst6, 2.0 ** -22)
    if local_scale == 1.0:
        assert worst < LOCAL_TOL


@pytest.mark.parametrize("seed,local_scale", [(2, 1.0), (7, 1.0), (11, 1.0), (2, 4.0)])
def test_occupancies_within_the_stated_bound(dev, seed, local_scale):
    """SconeOcc.forward on the inputs (clouds, queries, harmonics, captured hidden draws) of scone_occ.npz: variant 7 against the fp64
    oracle for every weight set; for the golden's own weights (seed 2) also against the REFERENCE's fp32 output.  Variant 6 on the same
    inputs is the control: fp32-class (1e-5)."""
    from macarons_amd import ops
    m, sd = _occ(dev, seed, local_scale)
    g = golden("scone_occ")
    worst, worst6 = 0.0, 0.0
    for tag in ("m100_q17", "m1024_q300", "m4096_q512"):
        perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)) for i in range(3)]
        pc, x, vh = T(g[f"{tag}_pc"], dev), T(g[f"{tag}_x"], dev), T(g[f"{tag}_vh"], dev)
        ref = nets.scone_occ_forward(sd, g[f"{tag}_pc"], g[f"{tag}_x"], g[f"{tag}_vh"], [p.numpy() for p in perms], np.float64)
        with torch.no_grad():
            with ops.variant(7):
                y7 = m(pc, x, vh, perms=perms).cpu().numpy()
            with ops.variant(6):
                y6 = m(pc, x, vh, perms=perms).cpu().numpy()
        assert y7.shape == ref.shape and np.isfinite(y7).all()
        worst, worst6 = max(worst, rel_err(y7, ref)), max(worst6, rel_err(y6, ref))
        if seed == 2 and local_scale == 1.0:
            assert rel_err(y7, g[f"{tag}_y"]) < OCC_TOL                      # the reference's own run
    _report(f"occupancy_rel_err_seed{seed}_x{local_scale:g}", {"variant7": worst, "variant6": worst6})
    assert worst < AMPLIFICATION * max(worst6, 2.0 ** -22)
    assert worst < OCC_TOL_BY_WEIGHTS[(seed, local_scale)] and worst6 < 1e-5
    assert m._full_range is False                                            # (the range guard stayed quiet: nothing fell back to variant 5)


@pytest.mark.parametrize("seed", [1, 5])
def test_scone_vis_single_plane(dev, seed):
    """SconeVis.forward at 2048 and 700 points (>= 512: the planes encoders, single-plane attention) on variant 7 against the fp64
    oracle; a batch gives every cloud the bits of its single call (launch-shape independence holds on this variant too); short clouds
    (< 512 points: the fp32-class kernels of every variant) stay at 1e-5."""
    from macarons_amd import ops
    from macarons_amd.networks import SconeVis
    with contextlib.redirect_stdout(io.StringIO()):
        m = SconeVis()
    sd = weights.make_state_dict(weights.shapes_of(m), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    rng = np.random.default_rng(70 + seed)
    worst = 0.0
    for B, N in ((1, 2048), (3, 700), (2, 100)):
        pts = np.concatenate([rng.uniform(-.5, .5, (B, N, 3)), rng.uniform(.1, 1., (B, N, 1))], -1).astype(np.float32)
        vh = (rng.standard_normal((B, N, 64)) * 0.3).astype(np.float32)
        with ops.variant(7), torch.no_grad():
            y = m(T(pts, dev), view_harmonics=T(vh, dev))
            for b in range(B):
                assert torch.equal(y[b:b + 1], m(T(pts[b:b + 1], dev), view_harmonics=T(vh[b:b + 1], dev))), (B, N, b)
        ref = nets.scone_vis_forward(sd, pts, vh, np.float64)
        e = rel_err(y.cpu().numpy(), ref)
        assert np.isfinite(y.cpu().numpy()).all()
        if N < 512:
            assert e < 1e-5
        else:
            worst = max(worst, e)
    _report(f"scone_vis_rel_err_seed{seed}", worst)
    assert worst < VIS_TOL and m._full_range is False


def test_variant7_is_deterministic_and_launch_shape_independent(dev):
    """A query's occupancy on variant 7 does not depend on how many queries share the launch (the rule of every variant: query shards
    of the multi-GPU step, chunks and the single call agree): chunks == whole, bit for bit; and a repeat returns the same bits."""
    from macarons_amd import ops
    m, sd = _occ(dev, 2)
    g = golden("scone_occ")
    tag = "m4096_q512"
    perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)) for i in range(3)]
    pc, x, vh = T(g[f"{tag}_pc"], dev), T(g[f"{tag}_x"], dev), T(g[f"{tag}_vh"], dev)
    with ops.variant(7), torch.no_grad():
        whole = m(pc, x, vh, perms=perms)
        again = m(pc, x, vh, perms=perms)
        parts = torch.cat([m(pc, x[:, a:b].contiguous(), vh[:, a:b].contiguous(), perms=perms) for a, b in ((0, 130), (130, 131), (131, 512))], 1)
    assert torch.equal(whole, again) and torch.equal(whole, parts)


@pytest.mark.parametrize("name", ["e2e_grid_config1", "e2e_grid_config2"])
def test_same_decision_as_the_reference_on_the_grid_goldens(dev, name):
    """BASELINE configs 1 and 2 (grid goldens of the reference): on variant 7 the arg-max camera must be the reference's; occupancies
    and gains inside the variant's bound; the sampled set is compared sample by sample and the differences are reported."""
    from test_nbv_gpu import run_end_to_end_on_grid
    r, g, rep, _ = run_end_to_end_on_grid(dev, name, variant=7)
    _report(name, rep)
    assert rep["fallback_variant"] is None
    assert rep["occ_rel_err"] < OCC_TOL and rep["gains_rel_err"] < GAIN_E2E_TOL
    assert rep["nbv_idx"] == rep["nbv_idx_ref"]
    # the sampled set: a few samples land on a neighbouring point of the CDF
    assert abs(rep["n_unique"] - rep["n_unique_ref"]) <= 0.02 * rep["n_unique_ref"], rep
    assert rep["samples_on_another_point"] <= 0.15 * rep["n_samples"] and rep["unique_points_in_common"] >= 0.85 * rep["n_unique_ref"], rep
    # ... and on the REFERENCE's sampled set the visibility network + scorer of this variant reproduce the reference's gains
    from macarons_amd import ops
    from macarons_amd.utility import scone_utils as su
    from macarons_amd.nbv import ViewStateGrid
    from test_nbv_gpu import _models
    occ, vis, _, _ = _models(dev)
    grid = ViewStateGrid(dev)
    pts = T(g["proxy"], dev)
    with ops.variant(7), torch.no_grad():
        vs = su.compute_view_state(pts[None, :, :3].contiguous(), T(g["X_view"], dev), grid.n_elev, grid.n_azim)
        vh = su.compute_view_harmonics(vs, grid.base_harmonics, grid.h_polar, grid.h_azim, grid.n_elev, grid.n_azim)
        harm = vis(pts[None].contiguous(), view_harmonics=vh.view(1, len(pts), 64))
        si = torch.from_numpy(g["sample_idx"].astype(np.int64)).to(dev)
        gains = vis.compute_coverage_gain(pts[si][None].contiguous(), harm[0][si][None].contiguous(), T(g["X_cam"], dev)[None])
    same_set = rel_err(gains.cpu().numpy().reshape(-1), g["gains"])
    _report(name + "_gains_on_reference_sample", same_set)
    assert same_set < GAIN_TOL


def test_same_ten_choices_as_the_reference_on_the_trajectory(dev):
    """BASELINE config 5's golden: ten consecutive MACARONS decisions of the reference, replayed on variant 7.  The same ten choices;
    per step the differing mask bits / value errors are reported, and bounded: frustum masks do not involve the networks (exact), the
    occupancy-driven state may move by single points."""
    from test_macarons_regime_gpu import run_macarons_trajectory
    rep = run_macarons_trajectory(dev, variant=7, strict=False)
    _report("macarons_trajectory", rep)
    assert [s["next_idx"] for s in rep["steps"]] == [s["next_idx_ref"] for s in rep["steps"]], rep["steps"]
    assert rep["draws_missing"] == []
    for s in rep["steps"]:
        assert s["fov_mask_bits"] == 0 and s["oof_bits"] == 0 and s["vs_rowsum_diff"] == 0, s        # geometry: no network in it
        assert s["field_rows"] == s["field_rows_ref"], s
        assert s["occ_rel_err"] is not None and s["occ_rel_err"] < OCC_TOL and s["gains_rel_err"] < GAIN_E2E_TOL, s
    assert rep["view_state_bits_final"] == 0 and rep["proxy_proba_final_rel_err"] < OCC_TOL


def test_headline_step_decision_equals_the_default_numerics(dev):
    """Q = 100k proxy points, M = 10 240, C = 200 (the size bench.py times as nbv_step): the variant-7 step picks the camera the default
    (variant 6) step picks, occupancies / gains inside the bound; bounded oracle sample at this size as for the default numerics."""
    from macarons_amd import ops
    from macarons_amd.nbv import nbv_step, ViewStateGrid
    from test_nbv_gpu import _models
    occ, vis, sdo, sdv = _models(dev)
    gen = torch.Generator().manual_seed(0)
    pc = (torch.rand(1, 10240, 3, generator=gen) - 0.5).to(dev)
    X = (torch.rand(1, 100_000, 3, generator=gen) - 0.5).to(dev)
    cams = torch.randn(200, 3, generator=gen)
    cams = (1.5 * cams / cams.norm(dim=1, keepdim=True)).to(dev)
    grid = ViewStateGrid(dev)
    torch.manual_seed(1)
    perms = occ.draw_perms(10240)
    u = torch.rand(2048, generator=gen).to(dev)
    out = {}
    for v in (6, 7):
        with ops.variant(v):
            out[v] = nbv_step(occ, vis, pc, X, cams[:3].contiguous(), cams, grid, occ_perms=perms, samples=u, return_samples=True)
    a, b = out[6], out[7]
    rep = {"occ_rel_diff": float((a["occ"] - b["occ"]).abs().max() / a["occ"].abs().max()),
           "gains_rel_diff": float((a["gains"] - b["gains"]).abs().max() / a["gains"].abs().max()),
           "nbv_idx": [int(a["nbv_idx"]), int(b["nbv_idx"])]}
    _report("headline_step_vs_variant6", rep)
    assert "fallback_variant" not in b and torch.isfinite(b["occ"]).all() and torch.isfinite(b["gains"]).all()
    assert rep["occ_rel_diff"] < OCC_TOL and rep["gains_rel_diff"] < GAIN_E2E_TOL and int(a["nbv_idx"]) == int(b["nbv_idx"])


def test_scene_batch_config3_on_variant_7(dev):
    """BASELINE config 3's shape at a size the suite affords (3 objects): the batched step on variant 7 == the single-cloud steps on
    variant 7 bit for bit (a cloud's numerics do not depend on the batch), and each decision == the default numerics' decision."""
    from macarons_amd import ops
    from macarons_amd.nbv import nbv_step, nbv_step_batch, draw_batch, ViewStateGrid
    from test_nbv_gpu import _models, _batch_scene
    occ, vis, sdo, sdv = _models(dev)
    B, M, Q, C = 3, 1024, 2048, 20
    pc, X, Xv, cams = _batch_scene(dev, B, M, Q, C, seed=14)
    torch.manual_seed(5)
    perms, u = draw_batch(occ, B, M, 2048, dev)
    grid = ViewStateGrid(dev)
    with ops.variant(7):
        rb = nbv_step_batch(occ, vis, pc, X, Xv, cams, grid, occ_perms=perms, samples=u)
        singles = [nbv_step(occ, vis, pc[b:b + 1], X[b:b + 1], Xv[b], cams, grid, occ_perms=[p[b] for p in perms], samples=u[b]) for b in range(B)]
    r6 = nbv_step_batch(occ, vis, pc, X, Xv, cams, grid, occ_perms=perms, samples=u)
    assert "fallback_variant" not in rb
    for b in range(B):
        assert torch.equal(rb["occ"][b], singles[b]["occ"]) and torch.equal(rb["gains"][b], singles[b]["gains"]), b
    assert torch.equal(rb["nbv_idx"].view(-1).cpu(), r6["nbv_idx"].view(-1).cpu())
    assert float((rb["occ"] - r6["occ"]).abs().max() / r6["occ"].abs().max()) < OCC_TOL


def test_sharded_step_on_variant_7_through_rccl_single_rank(dev, monkeypatch):
    """The variant is a property of the calling thread's calls, so a sharded step (rank-0 draws broadcast, occupancy all-gather, record
    merge -- run through RCCL on a one-rank group, MCR_FORCE_DIST_PATH) inside `ops.variant(7)` runs every network on variant 7 and
    reproduces the local variant-7 step bit for bit; the emulated share of an 8-rank step (bench.py: one_rank_of_8) runs on it too."""
    import socket
    import torch.distributed as dist
    from macarons_amd import ops
    from macarons_amd.nbv import nbv_step, nbv_step_one_rank_of, ViewStateGrid
    from test_nbv_gpu import _models
    g = golden("e2e_grid_config1")
    occ, vis, _, _ = _models(dev)
    grid = ViewStateGrid(dev)
    args = (occ, vis, T(g["pc"], dev), T(g["X"], dev), T(g["X_view"], dev), T(g["X_cam"], dev), grid)
    perms = [torch.from_numpy(g[f"perm{i}"].astype(np.int64)) for i in range(3)]
    with ops.variant(7):
        a = nbv_step(*args, occ_perms=perms, samples=T(g["samples"], dev))
    with ops.variant(6):
        a6 = nbv_step(*args, occ_perms=perms, samples=T(g["samples"], dev))
    assert not torch.equal(a["occ"], a6["occ"])                            # (variant 7 did run)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    monkeypatch.setenv("MCR_FORCE_DIST_PATH", "1")
    try:
        with ops.variant(7):
            b = nbv_step(*args, occ_perms=[p.to(dev) for p in perms], samples=T(g["samples"], dev), group=dist.group.WORLD)
        assert torch.equal(a["occ"], b["occ"]) and torch.equal(a["gains"], b["gains"]) and int(a["nbv_idx"]) == int(b["nbv_idx"])
    finally:
        dist.destroy_process_group()
    with ops.variant(7):
        e = nbv_step_one_rank_of(8, *args, [p.to(dev) for p in perms], T(g["samples"], dev))
    assert torch.isfinite(e["gains"]).all() and e["gains"].numel() >= 1


def test_ragged_pass_on_variant_7_equals_job_by_job(dev):
    """SconeOcc.forward_ragged (the occupancy-field pass of a MACARONS decision: J clouds / query chunks of different sizes in one launch
    sequence, incl. clouds of fewer than 2048 points -- padded global down-samples -- and a 1-row job) on variant 7 against the J
    forward() calls on variant 7 with the same draws: BIT-EQUAL for clouds of >= 512 points (the same kernels on the same rows); a
    cloud of fewer than 512 points runs its global transformer on the fp32-class short-sequence kernels when called alone and on the
    single-plane encoders inside the padded batch -- two different matrix paths, so those jobs agree within the variant's bound (variant 6
    has the same split at 1e-6); and everything within the bound of the fp64 oracle."""
    from macarons_amd import ops
    m, sd = _occ(dev, 2)
    rng = np.random.default_rng(12)
    sizes_m, sizes_q = [100, 3000, 65, 2048, 900], [17, 300, 1, 129, 4097]
    clouds = [rng.uniform(-.4, .4, (n, 3)).astype(np.float32) for n in sizes_m]
    xs = [rng.uniform(-.5, .5, (q, 3)).astype(np.float32) for q in sizes_q]
    vhs = [(rng.standard_normal((q, 64)) * .3).astype(np.float32) for q in sizes_q]
    torch.manual_seed(21)
    perms = [m.draw_perms(n) for n in sizes_m]
    with ops.variant(7), torch.no_grad():
        y = m.forward_ragged(T(np.concatenate(clouds), dev), sizes_m, T(np.concatenate(xs), dev), T(np.concatenate(vhs), dev), sizes_q,
                             perms=perms).cpu().numpy()
        refs = [m(T(c[None], dev), T(x[None], dev), T(v[None], dev), perms=p).cpu().numpy().reshape(-1, 1)
                for c, x, v, p in zip(clouds, xs, vhs, perms)]
    assert y.shape == (sum(sizes_q), 1) and np.isfinite(y).all()
    o = 0
    for M, r in zip(sizes_m, refs):
        if M >= 512:
            assert np.array_equal(y[o:o + len(r)], r), M
        else:
            assert rel_err(y[o:o + len(r)], r) < OCC_TOL, M
        o += len(r)
    o64 = np.concatenate([nets.scone_occ_forward(sd, c[None], x[None], v[None], [q.numpy() for q in p], np.float64).reshape(-1, 1)
                          for c, x, v, p in zip(clouds[:3], xs[:3], vhs[:3], perms[:3])])
    e = rel_err(y[:len(o64)], o64)
    _report("ragged_pass_rel_err", e)
    assert e < OCC_TOL
