"""Stage timings of one MACARONS decision on the bench scene (bench.measure_macarons_step's scene): where the host loop goes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd.utility import macarons_utils as mu

dev = torch.device("cuda:0")
marks = []
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); marks.append((name, time.perf_counter() - t0))
        return r
    setattr(mod, name, g)
wrap(mu, "compute_scene_occupancy_probability_field")
wrap(mu, "predict_coverage_gain_for_cameras")
from macarons_amd.utility import scene as sc
wrap(sc.Scene, "fill_cells")
wrap(sc.Scene, "update_from_depth")
wrap(sc.Scene, "set_all_features_to_value")
from macarons_amd.networks import SconeOcc as _SconeOccCls           # (the package re-exports the class under the module's name)
wrap(_SconeOccCls, "forward_ragged")
wrap(_SconeOccCls, "draw_perms")
from macarons_amd import ops as _ops
wrap(_ops, "scone_occ_forward_ragged")
wrap(_ops, "scone_vis_forward")
wrap(_ops, "sample_proxy_batched")
r = bench.measure_macarons_step(dev)
print(r["p50_ms"], r["last"])
import collections
agg = collections.defaultdict(list)
for k, v in marks: agg[k].append(v)
for k, v in agg.items(): print(f"{k:50s} n={len(v):5d}  median {sorted(v)[len(v)//2]*1e3:8.3f} ms  total/decision {sum(v)/11*1e3:8.2f} ms")
