"""Dev: per-phase cycle stamps of one workgroup of the fused local transformer (lib built with -DL6_TRACE=<block>)."""
import sys, os, torch, io, contextlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops, _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd.networks import SconeOcc
from macarons_amd.networks.packing import pack_local_pct
dev = torch.device("cuda:0")
v = int(os.environ.get("VARIANT", 6))
L = _lib.lib(); L.mcr_set_local_pct_variant(ctypes.c_int(v))
with contextlib.redirect_stdout(io.StringIO()):
    occ = SconeOcc().to(dev)
blob = pack_local_pct(occ.local_transformers[0], v)
if os.environ.get("ZERO_BLOB"): blob = torch.zeros_like(blob)
offs = torch.randn(16384, 16, 3, device=dev) * 0.05
names = ["emb1 product", "emb1 gelu+barrier", "emb2 gemm"]
for e in range(2):
    names += [f"e{e} prev epilogue", f"e{e} norm1 partial+barrier", f"e{e} norm1 finish+barrier", f"e{e} qkv gemm", f"e{e} barrier", f"e{e} qkv put+barrier",
              f"e{e} attention", f"e{e} bias+barrier", f"e{e} out gemm", f"e{e} res+norm2 partial+barrier", f"e{e} norm2 finish+barrier", f"e{e} ff1a gemm",
              f"e{e} ff1b gemm + gelu a", f"e{e} barrier", f"e{e} ff2a gemm + gelu b", f"e{e} barrier", f"e{e} (spare)", f"e{e} ff2b gemm"]
names += ["epilogue", "final norm+lin0", "pool"]
acc = None
for it in range(5):
    ops.local_pct_forward(offs, blob); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)(); L.mcr_dev_read_trace(buf)
    d = [buf[i + 1] - buf[i] for i in range(len(names))]
    acc = d if acc is None else [a + b for a, b in zip(acc, d)]
tot = sum(acc) / 5
print(f"last launch: {buf[len(names)] - buf[0]} shader ticks in {(buf[63] - buf[62]) * 10} ns (100 MHz wall clock) -> {(buf[len(names)] - buf[0]) / ((buf[63] - buf[62]) * 10):.2f} GHz")
print(f"total {tot:.0f} ticks")
for n, a in zip(names, acc): print(f"  {n:22s} {a/5:9.0f}  {100*a/5/tot:5.1f} %")
