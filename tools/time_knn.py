"""Dev: kNN kernel timing at the three scales of an NBV step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
Q = int(os.environ.get("Q", 100_000))
X = (torch.rand(1, Q, 3, generator=g) - 0.5).to(dev)
for M in [int(m) for m in os.environ.get("MS", "10240,1137,126").split(",")]:
    d = torch.randn(M, 3, generator=g); pc = (d / d.norm(dim=1, keepdim=True) * 0.3)[None].to(dev)      # a shell (surface-like)
    if os.environ.get("CLOUD") == "cube": pc = (torch.rand(1, M, 3, generator=g) - 0.5).to(dev)
    for _ in range(3): r = ops.knn_points(X, pc, 16, subtract_query=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = ops.knn_points(X, pc, 16, subtract_query=True)
    e1.record(); torch.cuda.synchronize()
    print(f"[{os.environ.get('MCR_DEV_LIB','main')} grid={os.environ.get('MCR_KNN_GRID','1')} {os.environ.get('CLOUD','shell')}] Q={Q} M={M}: {e0.elapsed_time(e1)/10*1e3:.1f} us  checksum {int(r[0].sum())}")
    if os.environ.get("KG_DEBUG"):
        import ctypes, numpy as np
        out = (ctypes.c_uint * (8192 * 16))()
        r = ops.knn_points(X, pc, 16, subtract_query=True); torch.cuda.synchronize()
        _lib.lib().mcr_knn_grid_debug(out)
        t = np.frombuffer(out, dtype=np.uint32).reshape(8192, 16)[: (Q + 31) // 32].astype(np.float64)
        life = t[:, 8:13].sum(1)
        print("   per wave  mean / p50 / p99 / max:")
        for name, v in (("sub-tiles", t[:, 0]), ("insert rounds", t[:, 1]), ("flushes", t[:, 2]), ("max-lane inserts", t[:, 3]), ("mean-lane inserts", t[:, 4]), ("cyc setup", t[:, 8]), ("cyc lb pass", t[:, 9]),
                        ("cyc seed", t[:, 10]), ("cyc rounds", t[:, 11]), ("cyc merge+out", t[:, 12]), ("cyc lifetime", life)):
            print("     %-18s %10.0f %10.0f %10.0f %10.0f" % (name, v.mean(), np.percentile(v, 50), np.percentile(v, 99), v.max()))
        hv = np.argsort(-t[:, 1])[:5]
        print("   heaviest waves (tiles, rounds, flushes, max-lane, mean-lane):", [tuple(int(v) for v in t[i, :5]) for i in hv])
