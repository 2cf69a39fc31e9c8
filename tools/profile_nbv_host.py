"""Dev: host-side (Python) time of nbv_step under cProfile."""
import sys, os, cProfile, pstats, io, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = argparse.Namespace(cams=200, nbv_iters=30)
pr = cProfile.Profile()
orig = bench.nbv_step if hasattr(bench, "nbv_step") else None
import macarons_amd.nbv as nbv
real = nbv.nbv_step
cnt = [0]
def wrapped(*a, **k):
    cnt[0] += 1
    if cnt[0] > 10:
        pr.enable()
    r = real(*a, **k)
    int(r["nbv_idx"])
    pr.disable()
    return r
nbv.nbv_step = wrapped
print(bench.measure_nbv_step(torch.device("cuda:0"), 0, 1, args)["p50_ms"])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
