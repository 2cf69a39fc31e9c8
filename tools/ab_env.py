"""A/B of one environment knob on ONE box (the boxes of the pool differ by more than most effects): alternates KNOB=0 / unset N times.
    python tools/ab_env.py MCR_OCC_X_EARLY nbv        (legs: nbv = headline NBV step p50, batch = config 3, mac = config 5)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
knob, leg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nbv")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
code = {
    "nbv": "import torch, bench, argparse; a = argparse.Namespace(cams=200, nbv_iters=60); r = bench.measure_nbv_step(torch.device('cuda:0'), 0, 1, a); print('RES', r['p50_ms'])",
    "batch": "import torch, bench, argparse; a = argparse.Namespace(cams=200, nbv_iters=60); r = bench.measure_nbv_batch(torch.device('cuda:0'), 0, 1, a); print('RES', r['p50_ms'])",
    "mac": "import torch, bench; r = bench.measure_macarons_step(torch.device('cuda:0')); print('RES', r['p50_ms'], r['variant_7']['p50_ms'])",
}[leg]
for rep in range(reps):
    for off in ("0", None):
        env = dict(os.environ)
        env.pop(knob, None)
        if off is not None:
            env[knob] = off
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600).stdout
        res = [ln for ln in out.splitlines() if ln.startswith("RES")]
        print(f"{knob}={'0' if off else 'default'}:", res[-1] if res else out[-300:], flush=True)
