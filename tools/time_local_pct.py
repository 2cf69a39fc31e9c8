import sys, os, torch, io, contextlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops, _lib
from macarons_amd.networks import SconeOcc
from macarons_amd.networks.packing import pack_local_pct
dev = torch.device("cuda:0")
v = int(os.environ.get("VARIANT", 1))
with contextlib.redirect_stdout(io.StringIO()):
    occ = SconeOcc().to(dev)
blob = pack_local_pct(occ.local_transformers[0], v)
S = int(os.environ.get("S", 16384))
offs = torch.randn(S, 16, 3, device=dev) * 0.05
with ops.variant(v):                      # (per-call selection: variant 7 is never a process default)
    for _ in range(int(os.environ.get("REP", 6))): ops.local_pct_forward(offs, blob)
torch.cuda.synchronize()
