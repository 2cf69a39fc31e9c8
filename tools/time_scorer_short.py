"""Dev: where the wall time of a 20-step scorer run goes (the driver's --steps 20 --warmup 5): host launch time, device time between
events, wake-up latency of the final synchronisation."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd import ops
import macarons_amd.torch_ops  # noqa
dev = torch.device("cuda:0")
pts, harm, cams = bench.make_inputs(100_000, 200, 1234, dev)
score = lambda: torch.ops.macarons.sh_coverage_gain(pts, harm, cams, True)
def step():
    return ops.best_record(score())
for _ in range(1000): step()
torch.cuda.synchronize()
for trial in range(6):
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(20):
        step(); ts.append(time.perf_counter())
    ev1.record()
    t1 = time.perf_counter()
    while not ev1.query(): pass
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    d = [(b - a) * 1e6 for a, b in zip([t0] + ts[:-1], ts)]
    print(f"trial {trial}: host launches {1e6*(t1-t0):7.1f} us (first step {d[0]:.1f}, median {sorted(d)[10]:.1f}, max {max(d):.1f}); "
          f"event fired at {1e6*(t2-t0):7.1f}; sync returned {1e6*(t3-t0):7.1f}; device between events {1e3*ev0.elapsed_time(ev1):7.1f} us")
