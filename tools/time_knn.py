"""Dev: kNN kernel timing at the three scales of an NBV step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
Q = 100_000
X = (torch.rand(1, Q, 3, generator=g) - 0.5).to(dev)
for M in [int(m) for m in os.environ.get("MS", "10240,1137,126").split(",")]:
    d = torch.randn(M, 3, generator=g); pc = (d / d.norm(dim=1, keepdim=True) * 0.3)[None].to(dev)
    for _ in range(3): r = ops.knn_points(X, pc, 16, subtract_query=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = ops.knn_points(X, pc, 16, subtract_query=True)
    e1.record(); torch.cuda.synchronize()
    print(f"[{os.environ.get('MCR_DEV_LIB','main')}] Q={Q} M={M}: {e0.elapsed_time(e1)/10*1e3:.1f} us  checksum {int(r[0].sum())}")
