"""Dev: where does variant 7 lose accuracy on the 4x local-weights case? per (S, scale) errors of v6 / v7 vs the fp64 oracle."""
import os, sys, io, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import weights
from oracle import nets
from macarons_amd import ops, _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd.networks import SconeOcc
from macarons_amd.networks.packing import pack_local_pct
dev = torch.device("cuda:0")
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
for scale in (1.0, 2.0, 4.0):
    with contextlib.redirect_stdout(io.StringIO()):
        m = SconeOcc()
    sd = weights.make_state_dict(weights.shapes_of(m), 2)
    sd = {k: (v * np.float32(scale) if (k.startswith("local_transformers.") and k.endswith("weight") and v.ndim == 2) else v) for k, v in sd.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to(dev).eval()
    rng = np.random.default_rng(6)
    for S, sc_in in ((3, 0.5), (1001, 0.05)):
        offs = (rng.standard_normal((S, 16, 3)) * sc_in).astype(np.float32)
        for sc in range(3):
            ref = nets.pc_transformer(sd, f"local_transformers.{sc}.", offs, np.float64)
            e = {}
            for v in (6, 7):
                with ops.variant(v), torch.no_grad():
                    e[v] = rel(ops.local_pct_forward(torch.from_numpy(offs).to(dev), pack_local_pct(m.local_transformers[sc], v)).cpu().numpy(), ref)
            print(f"[{os.environ.get('MCR_DEV_LIB','main')}] x{scale:g} S={S} in*{sc_in} scale {sc}: v6 {e[6]:.2e} v7 {e[7]:.2e}  max|ref| {np.abs(ref).max():.3g}")
