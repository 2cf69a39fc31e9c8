"""Dev: p50 of the MACARONS decision on the bench scene for several values of MCR_FIELD_GROUPS (host draws overlapped with GPU work)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):      # experimental build from tools/build_variant.py
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
import bench
os.environ["MCR_BENCH_NO_CHECKS"] = "1"
dev = torch.device("cuda:0")
for g in sys.argv[1:] or ["1", "2", "3", "4", ""]:
    if g:
        os.environ["MCR_FIELD_GROUPS"] = g
    else:
        os.environ.pop("MCR_FIELD_GROUPS", None)
    r = [bench.measure_macarons_step(dev)["p50_ms"] for _ in range(3)]
    print(f"groups={g or 'auto'}: p50 {sorted(r)[1]:.2f} ms  (runs {', '.join(f'{x:.2f}' for x in r)})", flush=True)
