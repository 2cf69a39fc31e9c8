import sys, os, torch, io, contextlib, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import weights
from oracle import nets
from macarons_amd import ops, _lib
if os.environ.get("MCR_DEV_LIB"):      # experimental build from tools/build_variant.py
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd.networks import SconeOcc
from macarons_amd.networks.packing import pack_local_pct
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    occ = SconeOcc()
sd = weights.make_state_dict(weights.shapes_of(occ), 2)
occ.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
occ = occ.to(dev)
S = 16384
offs = torch.randn(S, 16, 3, device=dev) * 0.05
ref = nets.pc_transformer(sd, "local_transformers.0.", offs[:600].cpu().numpy(), np.float64)
variants = [int(v) for v in os.environ.get("VARIANTS", "1,5,6").split(",")]
for rnd in range(2):
    for v in variants:
        blob = pack_local_pct(occ.local_transformers[0], v)
        if os.environ.get("ZERO_BLOB"): blob = torch.zeros_like(blob)      # power experiment: same instruction stream on zero operands
        if os.environ.get("ZERO_INPUT"): offs = torch.zeros_like(offs)
        with ops.variant(v):               # (per-call selection: variant 7 is never a process default)
            for _ in range(3): y = ops.local_pct_forward(offs, blob)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): y = ops.local_pct_forward(offs, blob)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        err = float(np.abs(y[:600].cpu().numpy() - ref).max() / np.abs(ref).max())
        print(f"[{os.environ.get('MCR_DEV_LIB', 'main')}] variant {v}: {dt*1e3:.3f} ms  {S*8.0e6/dt/1e12:.1f} TFLOP/s-equivalent   rel err vs fp64 oracle {err:.2e}")
