"""SconeVis.forward on a batch of 30 clouds x 2048 tokens (the neighbour cameras of a MACARONS decision) and the global PCTransformer
on 41 clouds: device time; MCR_DEV_LIB selects an experimental library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd.networks import SconeVis, SconeOcc
dev = torch.device("cuda:0")
torch.manual_seed(0)
vis, occ = SconeVis().to(dev).eval(), SconeOcc().to(dev).eval()
if os.environ.get("GUARD_OFF"):
    vis.range_guard = occ.range_guard = "off"
pts = torch.rand(30, 2048, 4, device=dev)
vh = torch.randn(30, 2048, 64, device=dev) * 0.3
pc = torch.rand(41, 2048, 3, device=dev) - 0.5


def bench(f, n=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    y = vis(pts, view_harmonics=vh)
    print(f"[{os.environ.get('MCR_DEV_LIB', 'main')}] SconeVis 30 x 2048: {bench(lambda: vis(pts, view_harmonics=vh)):.3f} ms; "
          f"global PCT 41 x 2048: {bench(lambda: occ.global_transformer(pc)):.3f} ms; checksum {float(y.double().abs().mean()):.9f}")
