"""Dev: scorer step (gain + reduce + decision record) eager vs hipGraph replay."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops
import bench
dev = torch.device("cuda:0")
pts, harm, cams = bench.make_inputs(100_000, 200, 1234, dev, cam_offset=0, n_cam_total=200)
def step(): return ops.best_record(ops.sh_coverage_gain(pts, harm, cams))
def t(fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("eager  %.2f us" % t(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    rec = step()
g.replay(); torch.cuda.synchronize()
ref = step()
print("graph == eager:", torch.equal(rec, ref))
print("graph  %.2f us" % t(g.replay))
