"""Dev: the headline NBV step as a hipGraph replay (nbv.GraphedNbvStep) a few times, for rocprofv3 kernel traces:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python tools/run_nbv_graph_steps.py 30
    python tools/trace_gaps.py out/t_kernel_trace.csv"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd.nbv import GraphedNbvStep, ViewStateGrid
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
grid = ViewStateGrid(dev)
gs = GraphedNbvStep(occ, vis, pc, X, cams[:3].contiguous(), cams, grid)
ts = []
for it in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = gs()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("p50 ms", float(np.median(ts[5:])) * 1e3)
