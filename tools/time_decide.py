import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops
import bench
dev = torch.device("cuda:0")
pts, harm, cams = bench.make_inputs(100_000, 200, 1234, dev, cam_offset=0, n_cam_total=200)
def t(fn, n=2000):
    for _ in range(200): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print("gain only              %.2f us" % t(lambda: ops.sh_coverage_gain(pts, harm, cams)))
    print("gain + best_record     %.2f us" % t(lambda: ops.best_record(ops.sh_coverage_gain(pts, harm, cams))))
    print("gain + torch.max       %.2f us" % t(lambda: torch.max(ops.sh_coverage_gain(pts, harm, cams), dim=1)))
    print("decide (fused)         %.2f us" % t(lambda: ops.sh_coverage_gain_decide(pts, harm, cams)))
