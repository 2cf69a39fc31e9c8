"""Dev: host-side (Python) time of one MACARONS decision under cProfile (bench.measure_macarons_step's scene)."""
import sys, os, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from macarons_amd.utility import macarons_utils as mu
pr = cProfile.Profile()
real = mu.macarons_nbv_decision
cnt = [0]
def wrapped(*a, **k):
    cnt[0] += 1
    if cnt[0] > 3: pr.enable()
    r = real(*a, **k)
    pr.disable()
    return r
mu.macarons_nbv_decision = wrapped
print(bench.measure_macarons_step(torch.device("cuda:0"))["p50_ms"], "decisions profiled:", cnt[0] - 3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
