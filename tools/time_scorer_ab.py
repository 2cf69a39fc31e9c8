"""A/B timing of the scorer: the gain kernel alone (mcr_sh_coverage_gain_partials) and the whole step, N=100k x C=200.
MCR_DEV_LIB=NAME selects tools/_libs/libmacarons_hip_NAME.so.  Interleaves nothing: run it once per library on the same box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd import ops
import bench
dev = torch.device("cuda:0")
pts, harm, cams = bench.make_inputs(100_000, 200, 1235, dev)
wps = int(os.environ.get("WPS", "0"))
def timeit(f, n):
    for _ in range(200): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    ka = timeit(lambda: ops.sh_coverage_gain_partials(pts, harm, cams, True, wps), 1000)
    st = timeit(lambda: ops.sh_coverage_gain(pts, harm, cams, True, wps), 1000)
    print(f"[{os.environ.get('MCR_DEV_LIB', 'main')} wps={wps}] kernel alone {ka:.2f} us   step {st:.2f} us   checksum {float(ops.sh_coverage_gain(pts, harm, cams, True, wps).double().sum()):.9f}")
