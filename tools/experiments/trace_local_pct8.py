"""Dev: per-group cycle stamps of one wave of local_pct8 (lib built with -DL8_TRACE=<block>): work and barrier wait per group."""
import sys, os, torch, io, contextlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macarons_amd import ops, _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd.networks import SconeOcc
from macarons_amd.networks.packing import pack_local_pct
dev = torch.device("cuda:0")
L = _lib.lib(); L.mcr_set_local_pct_variant(ctypes.c_int(8))
with contextlib.redirect_stdout(io.StringIO()):
    occ = SconeOcc().to(dev)
blob = pack_local_pct(occ.local_transformers[0], 8)
offs = torch.randn(16384, 16, 3, device=dev) * 0.05
names = ["emb1", "emb2a", "emb2b"]
for e in range(2):
    names += [f"e{e} LN1+QKV A", f"e{e} QKV B", f"e{e} QKV C", f"e{e} attention+out a", f"e{e} out b", f"e{e} LN2+FF0"] + [f"e{e} FF{p}" for p in range(1, 9)]
names += ["LN+lin0 a"]
acc = None
n = len(names)
for it in range(5):
    ops.local_pct_forward(offs, blob); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 160)(); L.mcr_dev_read_trace8(buf)
    # stamps: [work done g0, barrier passed g0, work done g1, ...]
    work = [buf[0] - 0] + [buf[2 * i] - buf[2 * i - 1] for i in range(1, n)]
    wait = [buf[2 * i + 1] - buf[2 * i] for i in range(n)]
    d = list(zip(work, wait))
    acc = d if acc is None else [(a + c, b + e_) for (a, b), (c, e_) in zip(acc, d)]
tot_w = sum(a for a, _ in acc[1:]) / 5; tot_b = sum(b for _, b in acc) / 5
print(f"work {tot_w:.0f} cycles (without group 0), barrier+DMA wait {tot_b:.0f} cycles")
for nm, (a, b) in zip(names, acc):
    print(f"  {nm:22s} work {a/5:8.0f}   wait {b/5:7.0f}")
