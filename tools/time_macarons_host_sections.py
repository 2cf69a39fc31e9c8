"""Dev: host wall time of the sections of a MACARONS decision WITHOUT extra synchronisations (what the host thread spends where;
a section that waits for the GPU shows its wait)."""
import os, sys, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd.utility import macarons_utils as mu, scene as sc
from macarons_amd.networks import SconeOcc as _Occ
from macarons_amd import ops as _ops
marks = collections.defaultdict(list)
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks[name].append(time.perf_counter() - t0); return r
    setattr(mod, name, g)
for m, n in ((mu, "compute_scene_occupancy_probability_field"), (mu, "predict_coverage_gain_for_cameras"), (sc.Scene, "fill_cells"),
             (sc.Scene, "update_from_depth"), (sc.Scene, "set_all_features_to_value"), (_Occ, "forward_ragged"), (_Occ, "draw_perms"),
             (_ops, "scone_occ_forward_ragged"), (_ops, "scone_vis_forward"), (_ops, "sample_proxy_batched"), (_ops, "points_in_fov"),
             (mu, "macarons_nbv_decision"), (sc.Scene, "fill_cells_begin"), (sc.Scene, "fill_cells_end"), (mu, "_field_select"),
             (_Occ, "forward_ragged_begin"), (_Occ, "forward_ragged_finish"), (_ops, "field_build"), (_ops, "field_finish"),
             (_ops, "uniform_rows"), (_ops, "h2d"), (_ops, "camera_boxes"), (_ops, "best_record")):
    wrap(m, n)
r = bench.measure_macarons_step(torch.device("cuda:0"))
print("p50 ms", r["p50_ms"])
nd = len(marks["macarons_nbv_decision"])
for k, v in marks.items():
    per = round(len(v) / nd)
    tail = v[len(v) // 2:]                                                     # the second half of the run (no first-call effects)
    print(f"{k:46s} calls/decision {len(v)/nd:6.1f}   host ms/decision {sum(tail)/max(len(tail),1)*max(per,1)*1e3:8.2f}")
