"""Quick timing of the network forwards at NBV-step size (run on the GPU box)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd.networks import SconeVis, SconeOcc

dev = torch.device("cuda:0")
if os.environ.get("VARIANT"):
    import ctypes
    from macarons_amd import _lib
    _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(int(os.environ["VARIANT"])))
torch.manual_seed(0)
vis, occ = SconeVis().to(dev).eval(), SconeOcc().to(dev).eval()
Q, M = int(os.environ.get("Q", 100000)), int(os.environ.get("M", 10240))
pc = torch.rand(1, M, 3, device=dev) - 0.5
x = torch.rand(1, Q, 3, device=dev) - 0.5
vh = torch.randn(1, Q, 64, device=dev) * 0.3
pts = torch.rand(1, 2048, 4, device=dev)
vh2 = torch.randn(1, 2048, 64, device=dev) * 0.3


def bench(f, n=5, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    print(f"SconeVis.forward N=2048: {bench(lambda: vis(pts, view_harmonics=vh2), 20, 3):.3f} ms")
    print(f"SconeOcc.forward M={M} Q={Q}: {bench(lambda: occ(pc, x, vh)):.2f} ms   (algorithmic {26.5e6*Q/1e12:.2f} TFLOP)")
