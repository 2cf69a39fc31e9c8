"""SconeVis.forward / global PCTransformer at one cloud of 2048 tokens, with and without the small-problem GEMM (env MCR_LINEAR3S=0/1
is read once per process: run twice)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd.networks import SconeVis, SconeOcc
dev = torch.device("cuda:0")
torch.manual_seed(0)
vis, occ = SconeVis().to(dev).eval(), SconeOcc().to(dev).eval()
pts = torch.rand(1, 2048, 4, device=dev)
vh = torch.randn(1, 2048, 64, device=dev) * 0.3
pc = torch.rand(1, 2048, 3, device=dev) - 0.5


def bench(f, n=200, warm=20):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    print(f"MCR_LINEAR3S={os.environ.get('MCR_LINEAR3S', '1')}: SconeVis.forward 2048 tokens {bench(lambda: vis(pts, view_harmonics=vh)):.1f} us; "
          f"global PCTransformer 2048 tokens {bench(lambda: occ.global_transformer(pc)):.1f} us")
