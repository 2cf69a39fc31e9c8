"""Dev: timeline of the driver's 20-step scorer run as bench.py issues it (steps round-robin on two side streams): host time of every
step() call, device time at which every step's stream reached its end (events), when the closing poll saw the end."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import macarons_amd.torch_ops  # noqa
dev = torch.device("cuda:0")
pts, harm, cams = bench.make_inputs(100_000, 200, 1234, dev)
S = int(os.environ.get("STREAMS", "2"))
K = int(os.environ.get("STEPS", "20"))
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
n = [0]
def step():
    st = streams[n[0] % S]; n[0] += 1
    with torch.cuda.stream(st):
        return torch.ops.macarons.sh_coverage_gain_best(pts, harm, cams, True)
for _ in range(1000): step()
torch.cuda.synchronize()
PRE = os.environ.get("PRE", "")
if "sleep" in PRE:
    time.sleep(0.05)
if "event" in PRE:                      # a timing event recorded on the current (default) stream, outside the timed trials
    e = torch.cuda.Event(enable_timing=True); e.record(); torch.cuda.synchronize()
if "step" in PRE:                       # two steps and a synchronise, outside the timed trials
    step(); step(); torch.cuda.synchronize()
if "sevent" in PRE:
    for st in streams:
        e = torch.cuda.Event(enable_timing=True); e.record(st)
    torch.cuda.synchronize()
for trial in range(3):
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    evs, ts = [], []
    t0 = time.perf_counter()
    ev0.record()
    if os.environ.get("EV0_ON_STREAMS"):
        pass
    for i in range(K):
        step(); ts.append(time.perf_counter())
        if os.environ.get("PER_STEP_EVENTS", "1") == "1":
            e = torch.cuda.Event(enable_timing=True); e.record(streams[i % S]); evs.append(e)
    ends = []
    for st in streams:
        e = torch.cuda.Event(enable_timing=True); e.record(st); ends.append(e)
    t1 = time.perf_counter()
    while not all(e.query() for e in ends): pass
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    host = [(t - t0) * 1e6 for t in ts]
    devt = [1e3 * ev0.elapsed_time(e) for e in evs]
    print(f"trial {trial}: wall {1e6*(t3-t0):.0f} us ({1e6*(t3-t0)/K:.1f} per step); launches done {1e6*(t1-t0):.0f}; end seen {1e6*(t2-t0):.0f}; "
          f"device end {max(1e3*ev0.elapsed_time(e) for e in ends):.0f}")
    print("   host step returns:", " ".join(f"{h:.0f}" for h in host))
    if devt: print("   device step ends :", " ".join(f"{d:.0f}" for d in devt))
