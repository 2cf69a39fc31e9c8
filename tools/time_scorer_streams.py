"""Dev: the scorer step (100k points x 200 cameras, gains + decision record) issued round-robin on 1 / 2 / 3 streams: consecutive steps are
independent batches, so the reduce of one can run beside the gain kernel of the next."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import macarons_amd.torch_ops  # noqa: F401
dev = torch.device("cuda:0")
C = int(os.environ.get("CAMS", "200"))
pts, harm, cams = bench.make_inputs(100_000, C, 1234, dev)
f = lambda: torch.ops.macarons.sh_coverage_gain_best(pts, harm, cams, True)
for _ in range(1500):
    f()
torch.cuda.synchronize()
for ns in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    for K in (20, 2000):
        for s in streams:
            with torch.cuda.stream(s):
                for _ in range(50):
                    f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            with torch.cuda.stream(streams[i % ns]):
                out = f()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{ns} stream(s), {K} steps: {dt / K * 1e6:.2f} us per step  ->  {C * K / dt / 1e6:.3f} M evals/s")
