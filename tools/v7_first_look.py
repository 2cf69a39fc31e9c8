"""Dev: variant 7 (the opt-in 16-bit matrix path) -- errors against the fp64 oracle / goldens and timings beside variant 6.
    python tools/v7_first_look.py            (on the GPU box)"""
import os, sys, time, io, contextlib
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import weights
from oracle import nets
from macarons_amd import ops
from macarons_amd.networks import SconeOcc
from macarons_amd.networks.packing import pack_local_pct

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(b).max())


def mod(seed, scale_local=1.0):
    with contextlib.redirect_stdout(io.StringIO()):
        m = SconeOcc()
    sd = weights.make_state_dict(weights.shapes_of(m), seed)
    if scale_local != 1.0:
        for k in sd:
            if k.startswith("local_transformers") and k.endswith("weight") and sd[k].ndim == 2:
                sd[k] = sd[k] * np.float32(scale_local)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval(), sd


def ev_time(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


m, sd = mod(2)
rng = np.random.default_rng(4)
print("== fused local transformer vs fp64 oracle (rel max-norm)")
for S in (3, 1001):
    offs = (rng.standard_normal((S, 16, 3)) * 0.05).astype(np.float32)
    for sc in range(3):
        ref = nets.pc_transformer(sd, f"local_transformers.{sc}.", offs, np.float64)
        out = {}
        for v in (6, 7):
            with ops.variant(v), torch.no_grad():
                out[v] = ops.local_pct_forward(T(offs), pack_local_pct(m.local_transformers[sc], v)).cpu().numpy()
        print(f"  S={S} scale {sc}: v6 {rel(out[6], ref):.2e}  v7 {rel(out[7], ref):.2e}")

print("== SconeOcc.forward on scone_occ.npz (vs the reference's fp32 output)")
g = np.load(os.path.join(ROOT, "tests", "golden", "scone_occ.npz"))
for seed, sl in ((2, 1.0),):
    for tag in ("m100_q17", "m1024_q300", "m4096_q512"):
        perms = [torch.from_numpy(g[f"{tag}_perm{i}"].astype(np.int64)) for i in range(3)]
        pc, x, vh = T(g[f"{tag}_pc"]), T(g[f"{tag}_x"]), T(g[f"{tag}_vh"])
        for v in (6, 7):
            with ops.variant(v), torch.no_grad():
                y = m(pc, x, vh, perms=perms).cpu().numpy()
            print(f"  {tag} v{v}: {rel(y, g[f'{tag}_y']):.2e}   (finite: {np.isfinite(y).all()})")

print("== timings")
for S in (16384, 100_000):
    offs = torch.randn(S, 16, 3, device=dev) * 0.05
    for v in (6, 7):
        blob = pack_local_pct(m.local_transformers[0], v)
        with ops.variant(v):
            def f():
                with ops.variant(v):
                    ops.local_pct_forward(offs, blob)
            print(f"  local_pct{v} S={S}: {ev_time(f):.3f} ms")

Q, M = 100_000, 10_240
gen = torch.Generator(device="cpu").manual_seed(4321)
d = torch.randn(M, 3, generator=gen)
pc = (d / d.norm(dim=1, keepdim=True) * torch.tensor([0.35, 0.25, 0.3]) + 0.002 * torch.randn(M, 3, generator=gen))[None].to(dev)
X = (torch.rand(1, Q, 3, generator=gen) - 0.5).to(dev)
vh = (torch.randn(1, Q, 64, generator=gen) * 0.3).to(dev)
torch.manual_seed(11)
perms = m.draw_perms(M)
ys = {}
for v in (6, 7):
    def f():
        with ops.variant(v), torch.no_grad():
            return m(pc, X, vh, perms=perms)
    ys[v] = f().cpu().numpy()
    print(f"  SconeOcc.forward Q=100k M=10240 v{v}: {ev_time(f, 10, 3):.3f} ms")
print(f"  v7 vs v6 at Q=100k: rel {rel(ys[7], ys[6]):.2e}")
