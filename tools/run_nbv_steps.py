"""Dev: the headline NBV step (Q=100k, M=10240, C=200) a few times, for rocprofv3 traces:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python tools/run_nbv_steps.py 30
    python tools/step_breakdown.py out/t_kernel_trace.csv"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd.nbv import nbv_step, nbv_step_one_rank_of, ViewStateGrid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
occ, vis = bench.build_models(dev)
g = torch.Generator(device="cpu").manual_seed(4321)
Q, M, C = 100_000, 10_240, 200
d = torch.randn(M, 3, generator=g)
pc = (d / d.norm(dim=1, keepdim=True) * torch.tensor([0.35, 0.25, 0.3]) + 0.002 * torch.randn(M, 3, generator=g))[None].to(dev)
X = (torch.rand(1, Q, 3, generator=g) - 0.5).to(dev)
cams = torch.randn(C, 3, generator=g)
cams = (1.5 * cams / cams.norm(dim=1, keepdim=True)).to(dev)
u = torch.rand(2048, generator=g).to(dev)
grid = ViewStateGrid(dev)
torch.manual_seed(11)
perms = [p.to(dev) for p in occ.draw_perms(M)]
import contextlib
from macarons_amd import ops
vctx = ops.variant(int(os.environ["VARIANT"])) if os.environ.get("VARIANT") else contextlib.nullcontext()   # e.g. VARIANT=7: the opt-in 16-bit matrix path
vctx.__enter__()
ts = []
for it in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if os.environ.get("EMULATE_WORLD"):       # one rank's share of an EMULATE_WORLD-rank step (bench.py: nbv_step.one_rank_of_8)
        r = nbv_step_one_rank_of(int(os.environ["EMULATE_WORLD"]), occ, vis, pc, X, cams[:3].contiguous(), cams, grid, perms, u)
    else:
        r = nbv_step(occ, vis, pc, X, cams[:3].contiguous(), cams, grid, occ_perms=perms, samples=u)
    int(r["host"]["nbv_idx"][0]) if "host" in r else int(r["nbv_idx"]); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("p50 ms", float(np.median(ts[5:])) * 1e3)
