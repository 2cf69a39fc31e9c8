"""Dev: the (cell, chunk) job sizes of the bench scene's MACARONS decision (candidate clouds M_j, query rows Q_j of the ragged occupancy pass)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd.networks import SconeOcc as _Occ
orig = _Occ.forward_ragged_begin
def spy(self, pc, cloud_sizes, x, vh, query_sizes, **kw):
    m, q = np.asarray(cloud_sizes), np.asarray(query_sizes)
    print("J", len(m), "M: min %d med %d max %d sum %d | Q: min %d med %d max %d sum %d" % (m.min(), np.median(m), m.max(), m.sum(), q.min(), np.median(q), q.max(), q.sum()))
    print("M", sorted(m.tolist())); print("Q", sorted(q.tolist()))
    _Occ.forward_ragged_begin = orig
    return orig(self, pc, cloud_sizes, x, vh, query_sizes, **kw)
_Occ.forward_ragged_begin = spy
os.environ["MCR_BENCH_NO_CHECKS"] = "1"
r = bench.measure_macarons_step(torch.device("cuda:0"))
print(r["p50_ms"])
