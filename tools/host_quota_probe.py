"""Shows the host-thread throttling that macarons_amd.utility.host guards against: the container quota, torch's pool size and a
1 MB tensor copy timed with the default pool and with one thread (gpurun -- python tools/host_quota_probe.py)."""
import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "-", type(e).__name__)
a = torch.empty(1 << 18); b = torch.empty(1 << 18)
for n in (torch.get_num_threads(), 1):
    torch.set_num_threads(n)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); b.copy_(a); ts.append(time.perf_counter() - t0)
    ts.sort(); print("copy 1MB threads", n, "p50 %.3f ms max %.3f ms" % (ts[150] * 1e3, ts[-1] * 1e3))
