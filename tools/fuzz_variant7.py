"""Dev: shape fuzz of the opt-in 16-bit matrix path -- SconeOcc.forward and SconeVis.forward on variant 7 against variant 6 over
random (clouds, surface points, queries) incl. tile edges of every kernel involved (4-query workgroups of local_pct7, 64 / 128 / 256-row
GEMM tiles, 64-key attention tiles, the 512-token switch of the encoders): finite, within the variant's bound of variant 6 (max-norm with a floor of 1.0), chunked ==
whole bit for bit.   python tools/fuzz_variant7.py [n_cases]"""
import os, sys, io, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import weights
from macarons_amd import ops
from macarons_amd.networks import SconeOcc, SconeVis
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
with contextlib.redirect_stdout(io.StringIO()):
    occ, vis = SconeOcc(), SconeVis()
occ.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict(weights.shapes_of(occ), 2).items()})
vis.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict(weights.shapes_of(vis), 1).items()})
occ, vis = occ.to(dev).eval(), vis.to(dev).eval()
rng = np.random.default_rng(2026)
edges_q = [1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1025, 4095, 4097, 20001]
edges_m = [64, 65, 100, 511, 512, 513, 1024, 2047, 2048, 2049, 5000, 16384, 20000]
worst = {"occ": 0.0, "vis": 0.0}
for case in range(n_cases):
    B = int(rng.choice([1, 1, 1, 2, 3]))
    M, Q = int(rng.choice(edges_m)), int(rng.choice(edges_q))
    pc = torch.from_numpy(rng.uniform(-.4, .4, (B, M, 3)).astype(np.float32)).to(dev)
    x = torch.from_numpy(rng.uniform(-.5, .5, (B, Q, 3)).astype(np.float32)).to(dev)
    vh = torch.from_numpy((rng.standard_normal((B, Q, 64)) * .3).astype(np.float32)).to(dev)
    torch.manual_seed(case)
    perms = occ.draw_perms(M)
    with torch.no_grad():
        with ops.variant(7):
            y7 = occ(pc, x, vh, perms=perms)
            if Q > 5:
                cut = int(rng.integers(1, Q))
                parts = torch.cat([occ(pc, x[:, :cut].contiguous(), vh[:, :cut].contiguous(), perms=perms),
                                   occ(pc, x[:, cut:].contiguous(), vh[:, cut:].contiguous(), perms=perms)], 1)
                assert torch.equal(parts, y7), ("chunks != whole", B, M, Q, cut)
        with ops.variant(6):
            y6 = occ(pc, x, vh, perms=perms)
    assert torch.isfinite(y7).all(), ("non-finite", B, M, Q)
    e = float((y7 - y6).abs().max() / max(float(y6.abs().max()), 1.0))      # (max-norm; the floor = the scale of the occupancies of a full query set, 1 .. 3, for launches of a few queries)
    worst["occ"] = max(worst["occ"], e)
    assert e < 4e-3, ("occ", B, M, Q, e)
    N = int(rng.choice([16, 100, 511, 512, 513, 700, 1024, 2047, 2048]))
    pts = torch.from_numpy(np.concatenate([rng.uniform(-.5, .5, (B, N, 3)), rng.uniform(.1, 1., (B, N, 1))], -1).astype(np.float32)).to(dev)
    vhs = torch.from_numpy((rng.standard_normal((B, N, 64)) * .3).astype(np.float32)).to(dev)
    with torch.no_grad():
        with ops.variant(7):
            h7 = vis(pts, view_harmonics=vhs)
        with ops.variant(6):
            h6 = vis(pts, view_harmonics=vhs)
    assert torch.isfinite(h7).all(), ("vis non-finite", B, N)
    ev = float((h7 - h6).abs().max() / h6.abs().max())
    worst["vis"] = max(worst["vis"], ev)
    assert ev < 5e-3, ("vis", B, N, ev)
    print(f"case {case:3d}: B={B} M={M:6d} Q={Q:6d} occ {e:.2e}   N={N:5d} vis {ev:.2e}", flush=True)
print("worst", worst, "module fell back to variant 5:", occ._full_range, vis._full_range)
