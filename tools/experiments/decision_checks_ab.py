import os, sys, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device("cuda:0")
for chk in ("", "1", "", "1"):
    if chk: os.environ["MCR_BENCH_NO_CHECKS"] = "1"
    else: os.environ.pop("MCR_BENCH_NO_CHECKS", None)
    r = bench.measure_macarons_step(dev)
    print("no_checks" if chk else "checks   ", round(r["p50_ms"], 2), flush=True)
