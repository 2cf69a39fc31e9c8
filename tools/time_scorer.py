"""Dev: scorer step time for an experimental library (MCR_DEV_LIB) or the product library."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd import ops
import bench
dev = torch.device("cuda:0")
pts, harm, cams = bench.make_inputs(100_000, 200, 1234, dev, cam_offset=0, n_cam_total=200)
for wps in [int(x) for x in os.environ.get("WPS", "0").split(",")]:
    for _ in range(100): g = ops.sh_coverage_gain(pts, harm, cams, True, wps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): g = ops.sh_coverage_gain(pts, harm, cams, True, wps)
    e1.record(); torch.cuda.synchronize()
    print(f"[{os.environ.get('MCR_DEV_LIB','main')}] waves/SIMD={wps}: {e0.elapsed_time(e1):.2f} us/step  gain[0]={float(g[0,0]):.7f}")
