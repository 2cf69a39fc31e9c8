"""Dev: the 1344 -> 512 head layer (T = 100k rows) on linear3p through a SconeOcc forward slice: times the head kernels from a trace-free
event pair around SconeOcc.forward minus nothing -- use with rocprofv3 --kernel-trace, or MCR_L3P_KO knock-outs (128-tile kernel)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
import bench
dev = torch.device("cuda:0")
occ, vis = bench.build_models(dev)
g = torch.Generator().manual_seed(1)
Q, M = 100_000, 10240
pc = (torch.rand(1, M, 3, generator=g) - 0.5).to(dev)
X = (torch.rand(1, Q, 3, generator=g) - 0.5).to(dev)
vh = (torch.randn(1, Q, 64, generator=g) * 0.3).to(dev)
perms = [p.to(dev) for p in occ.draw_perms(M)]
occ.range_guard = "off"
for _ in range(3): occ(pc, X, vh, perms=perms)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): occ(pc, X, vh, perms=perms)
torch.cuda.synchronize(); print("SconeOcc forward ms", (time.perf_counter() - t0) / 10 * 1e3)
