"""Energy per query of the fused local transformer: variant 6 (fp16 hi/lo planes, three MFMAs per product, the default) against variant 7
(ONE fp16 plane, the opt-in 16-bit matrix path): socket power (hwmon of the GPU's PCI function) and shader clock while each kernel
runs back to back for ~3 s, three alternating rounds on ONE box.  J per query = mean W x seconds / queries; cycles per 16 384 queries =
mean sclk x ms.
    python tools/energy_local_pct7.py > gpurun_out/power_local_pct7.txt        (GPU box)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from power_trace import Sampler


def main():
    dev = torch.device("cuda:0")
    import contextlib, io
    import numpy as np
    from macarons_amd import ops
    from macarons_amd.networks import SconeOcc
    from macarons_amd.networks.packing import pack_local_pct
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights
    with contextlib.redirect_stdout(io.StringIO()):
        occ = SconeOcc()
    sd = weights.make_state_dict(weights.shapes_of(occ), 2)
    occ.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    occ = occ.to(dev)
    pct = occ.local_transformers[0]
    variants = [int(v) for v in os.environ.get("VARIANTS", "6,7").split(",")]
    blobs = {v: pack_local_pct(pct, v) for v in variants}
    Q = 16384 * 4
    offs = torch.randn(Q, 16, 3, device=dev) * 0.05
    s = Sampler()
    s.start()
    time.sleep(1.5)
    res = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            with ops.variant(v):
                for _ in range(20):
                    ops.local_pct_forward(offs, blobs[v])
                torch.cuda.synchronize()
                s.tag = f"v{v}_r{rnd}"
                t0 = time.perf_counter()
                n = 0
                while time.perf_counter() - t0 < 3.0:
                    for _ in range(20):
                        ops.local_pct_forward(offs, blobs[v])
                    torch.cuda.synchronize()
                    n += 20
                dt = time.perf_counter() - t0
            s.tag = "idle"
            res[v].append((n, dt))
            time.sleep(1.0)
    s.stop = True
    s.join()
    print(f"telemetry: {'hwmon ' + s.hw[0] if s.hw else 'rocm-smi'}; {len(s.rows)} samples")
    print(f"{'run':8s} {'ms/16384q':>10s} {'W mean':>8s} {'W max':>7s} {'sclk MHz':>9s} {'uJ/query':>9s} {'Mcycles/16384q':>15s}")
    agg = {}
    for v in variants:
        for rnd, (n, dt) in enumerate(res[v]):
            rows = [r for r in s.rows if r[1] == f"v{v}_r{rnd}"]
            rows = rows[len(rows) // 5:]                            # (the first fifth: power and clock still settling)
            w = float(np.mean([r[2] for r in rows])); wmax = float(np.max([r[2] for r in rows])); mhz = float(np.mean([r[3] for r in rows]))
            ms = dt / n / (Q / 16384) * 1e3
            uj = w * (dt / n) / Q * 1e6
            print(f"v{v} r{rnd}   {ms:10.4f} {w:8.1f} {wmax:7.1f} {mhz:9.1f} {uj:9.3f} {mhz * 1e6 * ms * 1e-3 / 1e6:15.3f}")
            agg.setdefault(v, []).append((ms, w, mhz, uj))
    for v in variants:
        a = np.array(agg[v])
        print(f"v{v} mean  {a[:, 0].mean():10.4f} {a[:, 1].mean():8.1f} {'':7s} {a[:, 2].mean():9.1f} {a[:, 3].mean():9.3f} {(a[:, 2] * a[:, 0]).mean() * 1e-3:15.3f}")
    idle = [r for r in s.rows if r[1] == "idle"]
    print(f"idle: {np.mean([r[2] for r in idle[:50]]):.1f} W")


main()
