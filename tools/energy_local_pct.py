"""Energy per query of the fused local transformer, v6 (shipped) against v8 (tools/experiments/local_pct8.hip, dev library): socket
power (hwmon of the GPU's PCI function, >= 40 Hz) and shader clock while each kernel runs back to back for ~4 s, three alternating
rounds on ONE box.  J per query = mean W x seconds / queries; cycles per 16 384 queries = mean sclk x ms.  Closes the round-3 question
"is there energy left to save in v8?" (VERDICT r3 next #4).
    python tools/build_variant.py v8 "networks.hip,+tools/experiments/local_pct8.hip" -DMCR_DEV_LOCAL_PCT8      (build container)
    python tools/energy_local_pct.py > gpurun_out/local_pct8_energy.txt                                         (GPU box)"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
from macarons_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_libs", "libmacarons_hip_v8.so")        # product kernels + variant 8
from power_trace import Sampler


def main():
    dev = torch.device("cuda:0")
    import contextlib, io
    import numpy as np
    from macarons_amd import ops
    from macarons_amd.networks import SconeOcc
    from macarons_amd.networks.packing import pack_local_pct
    from pack_local_pct8 import _pack_local_pct8
    L = _lib.lib()
    with contextlib.redirect_stdout(io.StringIO()):
        occ = SconeOcc().to(dev)
    pct = occ.local_transformers[0]
    blobs = {6: pack_local_pct(pct, 6), 8: _pack_local_pct8(pct)}
    Q = 16384 * 4
    offs = torch.randn(Q, 16, 3, device=dev) * 0.05
    out = {}
    for v in (6, 8):                                               # same bits class: compare the two kernels' outputs first
        L.mcr_set_local_pct_variant(ctypes.c_int(v))
        out[v] = ops.local_pct_forward(offs, blobs[v]).clone()
    torch.cuda.synchronize()
    print(f"max |v8 - v6| / max |v6| on {Q} queries: {float((out[8] - out[6]).abs().max() / out[6].abs().max()):.2e}")
    s = Sampler()
    s.start()
    time.sleep(1.5)
    res = {6: [], 8: []}
    for rnd in range(3):
        for v in (6, 8):
            L.mcr_set_local_pct_variant(ctypes.c_int(v))
            for _ in range(20):
                ops.local_pct_forward(offs, blobs[v])
            torch.cuda.synchronize()
            s.tag = f"v{v}_r{rnd}"
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 4.0:
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
    for v in (6, 8):
        for rnd, (n, dt) in enumerate(res[v]):
            rows = [r for r in s.rows if r[1] == f"v{v}_r{rnd}"]
            rows = rows[len(rows) // 5:]                            # (the first fifth: power and clock still settling)
            w = float(np.mean([r[2] for r in rows])); wmax = float(np.max([r[2] for r in rows])); mhz = float(np.mean([r[3] for r in rows]))
            ms = dt / n / (Q / 16384) * 1e3
            uj = w * (dt / n) / Q * 1e6
            print(f"v{v} r{rnd}   {ms:10.4f} {w:8.1f} {wmax:7.1f} {mhz:9.1f} {uj:9.3f} {mhz * 1e6 * ms * 1e-3 / 1e6:15.3f}")
            agg.setdefault(v, []).append((ms, w, mhz, uj))
    for v in (6, 8):
        a = np.array(agg[v])
        print(f"v{v} mean  {a[:, 0].mean():10.4f} {a[:, 1].mean():8.1f} {'':7s} {a[:, 2].mean():9.1f} {a[:, 3].mean():9.3f} {(a[:, 2] * a[:, 0]).mean() * 1e-3:15.3f}")
    idle = [r for r in s.rows if r[1] == "idle"]
    print(f"idle: {np.mean([r[2] for r in idle[:50]]):.1f} W")
    a6, a8 = np.array(agg[6]), np.array(agg[8])
    print(f"v8 / v6: time {a8[:, 0].mean() / a6[:, 0].mean():.3f}, power {a8[:, 1].mean() / a6[:, 1].mean():.3f}, energy per query "
          f"{a8[:, 3].mean() / a6[:, 3].mean():.3f}, clock {a8[:, 2].mean() / a6[:, 2].mean():.3f}")


if __name__ == "__main__":
    main()
