"""Dev: cProfile of the host side of MACARONS decisions on the bench scene (where the Python time of a decision goes)."""
import cProfile, pstats, os, sys, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
os.environ["MCR_BENCH_NO_CHECKS"] = "1"
dev = torch.device("cuda:0")
bench.measure_macarons_step(dev)           # warm: caches, arenas
pr = cProfile.Profile()
pr.enable()
r = bench.measure_macarons_step(dev)
pr.disable()
print("p50", r["p50_ms"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
print(s.getvalue()[:14000])
