"""Dev: host time of SconeOcc's hidden draws (torch.ops.macarons.scone_occ_draws) at the job sizes of the bench decision
(MCR_MT_SCALAR=1: the scalar state transition)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macarons_amd.torch_ops  # noqa: F401
rng = np.random.default_rng(0)
m0 = [int(x) for x in rng.integers(1000, 15000, 46)]
ds = lambda m: max(int((m / 128) ** 0.5), 1) or 2
m1 = [m // ds(m) for m in m0]
m2 = [a // ds(m) for a, m in zip(m1, m0)]
f = torch.ops.macarons.scone_occ_draws
for _ in range(5):
    f(m0, m1, m2, 2048)
t = time.perf_counter()
for _ in range(50):
    f(m0, m1, m2, 2048)
print(f"scone_occ_draws ({sum(m0)} cloud points, 46 jobs): {(time.perf_counter() - t) / 50 * 1e6:.1f} us  "
      f"[{'scalar' if os.environ.get('MCR_MT_SCALAR') else 'avx2'} transition]")
