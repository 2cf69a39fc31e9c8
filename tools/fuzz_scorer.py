"""Dev: the scorer (gains and per-point visibilities, sigmoid and relu) against the C port of the reference scorer (oracle/cport) on
random shapes: clouds of 1 .. 5000 points (tile edges: N = 1, 63, 64, 65, 127, 128, 129 ...), 1 .. 300 cameras, 1 .. 3 clouds, point
stride 3 or 4.  python tools/fuzz_scorer.py [cases]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd import ops
from oracle import cport, scorer as oscorer
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
rng = np.random.default_rng(2024)
edge_n = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 1000, 4095, 4096, 4097]
worst = 0.0
for case in range(n_cases):
    B = int(rng.integers(1, 4))
    N = int(edge_n[case % len(edge_n)]) if case % 2 == 0 else int(rng.integers(1, 5000))
    C = int(rng.choice([1, 2, 3, 7, 20, 52, 64, 100, 200, 300]))
    P = int(rng.choice([3, 4]))
    sig = bool(rng.integers(0, 2))
    pts = rng.uniform(-.5, .5, (B, N, P)).astype(np.float32)
    harm = (rng.standard_normal((B, N, 64)) * rng.choice([0.1, 0.5, 1.5])).astype(np.float32)
    cams = rng.standard_normal((B, C, 3)).astype(np.float32)
    cams = (1.5 * cams / np.linalg.norm(cams, axis=-1, keepdims=True)).astype(np.float32)
    ref, _ = cport.coverage_gain(pts, harm, cams, use_sigmoid=sig)
    t = lambda a: torch.from_numpy(a).to(dev)
    got = ops.sh_coverage_gain(t(pts), t(harm), t(cams), sig, 0).cpu().numpy()
    vis = ops.sh_visibilities(t(pts), t(harm), t(cams), sig).cpu().numpy()          # [B, C, N]
    err = np.abs(got - ref).max() / max(1e-6, np.abs(ref).max())
    err_v = np.abs(vis.mean(-1) - ref).max() / max(1e-6, np.abs(ref).max())
    # fp32 against fp32: a relu output of a small cloud is a raw sum of 64 cancelling terms, and two fp32 evaluations of it differ by
    # more than either differs from the truth -- past 2e-5 both are measured against the fp64 evaluation of the same formula
    if max(err, err_v) > 2e-5:
        truth = oscorer.compute_coverage_gain(pts[..., :3], harm, cams, use_sigmoid=sig, dtype=np.float64)
        scale = max(1e-6, np.abs(truth).max())
        e_ours, e_port = np.abs(got - truth).max() / scale, np.abs(ref - truth).max() / scale
        print(f"  case {case} B={B} N={N} C={C} sigmoid={sig}: vs C port {err:.2e}; vs fp64: ours {e_ours:.2e}, C port {e_port:.2e}")
        err = err_v = e_ours
    if max(err, err_v) > worst:
        worst, worst_case = max(err, err_v), f"B={B} N={N} C={C} P={P} sigmoid={sig}"
    if not (err < 1e-4 and err_v < 1e-4 and np.isfinite(got).all()):            # the path's tolerance (1e-4 relative)
        print(f"MISMATCH case {case}: B={B} N={N} C={C} P={P} sigmoid={sig}: gains {err:.2e}, mean of visibilities {err_v:.2e}")
        sys.exit(1)
print(f"[{os.environ.get('MCR_DEV_LIB', 'main')}] {n_cases} cases, worst relative deviation {worst:.2e} ({worst_case}; against the fp64 evaluation where the two fp32 results are more than 2e-5 apart)")
