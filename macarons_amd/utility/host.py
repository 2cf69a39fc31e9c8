"""Host CPU budget of the process.

PyTorch sizes its intra-op thread pool from the machine's core count (128 on the 2 x 64-core MI355X hosts) and ignores the
container's CPU quota (cgroup cpu.max: 16 CPUs on the pool's GPU boxes).  A parallel region of 128 threads then burns the
whole 100 ms CFS period's quota in a few milliseconds and the kernel THROTTLES every thread of the process, the one that
launches kernels included, for the rest of the period: a 1 MB `tensor.copy_` measured p50 0.02 ms / max 88 ms, and a MACARONS
decision whose host glue touches ~1 MB index arrays went from 16 ms to 96 ms.  `limit_host_threads()` caps torch's pool at
what the process may actually use; macarons_amd.ops applies it on import (MCR_HOST_THREADS=0 leaves torch alone,
MCR_HOST_THREADS=N asks for N).
"""
import torch
import math
import os


def _cgroup_quota():
    """CPUs the cgroup lets this process burn per period (None: unlimited or unknown)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                                  # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()[:2]
        if quota != "max" and int(period) > 0:
            return max(1, math.ceil(int(quota) / int(period)))
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:                     # cgroup v1
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            return max(1, math.ceil(quota / period))
    except (OSError, ValueError):
        pass
    return None


def effective_cpus():
    """CPUs this process can keep busy: the scheduler affinity, capped by the cgroup quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = _cgroup_quota()
    return max(1, min(n, quota) if quota else n)


def limit_host_threads(n=None):
    """Cap torch's intra-op pool at `n` (default: effective_cpus(), or MCR_HOST_THREADS).  Never raises the count.
    Returns the pool size in force afterwards."""
    import torch
    if n is None:
        env = os.environ.get("MCR_HOST_THREADS", "").strip()
        if env == "0":
            return torch.get_num_threads()
        n = int(env) if env else effective_cpus()
    n = max(1, int(n))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


# ---- the reference's hidden torch.randperm draws, batched ---------------------------------------------------------------------------
_TORCH_RANDPERM = torch.randperm


def batched_draws_ok():
    """True while torch.randperm is torch's own: the C++ operators that make a decision's ~200 hidden draws in two calls
    (torch.ops.macarons.randperm_prefixes / scone_occ_draws: the same at::randperm calls on the same generator) bypass the Python
    name -- a harness that replaced it (the golden generator's keyed draws, tests/golden/keyed_rng.py) gets the Python loop instead."""
    return torch.randperm is _TORCH_RANDPERM
