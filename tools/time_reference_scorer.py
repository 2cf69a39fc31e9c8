"""BUILD CONTAINER ONLY (the reference never travels): time the IMPORTED reference scorer, SconeVis.compute_coverage_gain
(/root/reference/macarons/networks/SconeVis.py:210-252), at the headline size N = 100 000 points x C = 200 cameras on this container's
host cores -- in camera chunks of 20 (one call over all 200 cameras would allocate ~15 GB of temporaries) -- and print the line that
BASELINE.md section 2 quotes.  Inputs: bench.py's make_inputs (seed 1234).
    python tools/time_reference_scorer.py [N] [C] [chunk]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import _ref_import
ref = _ref_import.load_reference()
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 200
CH = int(sys.argv[3]) if len(sys.argv) > 3 else 20
threads = os.cpu_count() or 1
torch.set_num_threads(threads)
g = torch.Generator().manual_seed(1234)
pts = torch.cat((torch.rand(1, N, 3, generator=g) - 0.5, torch.rand(1, N, 1, generator=g) * 0.9 + 0.1), -1)
harm = torch.randn(1, N, 64, generator=g) * 0.5
cams = torch.randn(1, C, 3, generator=g)
cams = 1.5 * cams / cams.norm(dim=-1, keepdim=True)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    vis = ref["SconeVis"].SconeVis()


def run():
    out = []
    with torch.no_grad():
        for c0 in range(0, C, CH):
            out.append(vis.compute_coverage_gain(pts, harm, cams[:, c0:c0 + CH]))
    return torch.cat(out, 1)


run()                                               # warm-up
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    gains = run()
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print(f"reference SconeVis.compute_coverage_gain, N={N}, C={C} in chunks of {CH}, fp32, torch {torch.__version__} CPU, {threads} threads: "
      f"{t:.2f} s per call (median of 3) = {C / t:.1f} evals/s = {N * C / t / 1e6:.2f} M point-camera pairs/s; gains [{float(gains.min()):.4f}, {float(gains.max()):.4f}]")
