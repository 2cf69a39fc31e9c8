import sys, os, torch, io, contextlib, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops, _lib
from macarons_amd.networks import SconeOcc
from macarons_amd.networks.packing import pack_local_pct
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    occ = SconeOcc().to(dev)
blob = pack_local_pct(occ.local_transformers[0])
S = 16384
offs = torch.randn(S, 16, 3, device=dev) * 0.05
outs = {}
for rnd in range(3):
    for v in (1, 2):
        _lib.lib().mcr_set_local_pct_variant(ctypes.c_int(v))
        for _ in range(3): y = ops.local_pct_forward(offs, blob)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): y = ops.local_pct_forward(offs, blob)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        outs[v] = y
        print(f"variant {v}: {dt*1e3:.3f} ms  {S*8.0e6/dt/1e12:.1f} TFLOP/s")
print("max rel diff v1 vs v2:", float((outs[1] - outs[2]).abs().max() / outs[1].abs().max()))
