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
             (_ops, "uniform_rows"), (_ops, "h2d"), (_ops, "camera_boxes"), (_ops, "best_record"), (mu, "_field_prepare"), (mu, "_job_groups"),
             (sc.Scene, "fill_cells_draw"), (sc.Scene, "fill_cells_apply"), (_ops, "field_select"), (_ops, "scene_fill_begin")):
    wrap(m, n)
r = bench.measure_macarons_step(torch.device("cuda:0"))
print("p50 ms", r["p50_ms"])
nd = len(marks["macarons_nbv_decision"])
half = nd // 2                                                                # the second half of the decisions (no first-call effects)
for k, v in marks.items():
    per = len(v) / nd
    n_tail = int(round(per * (nd - half)))
    tail = v[len(v) - n_tail:] if n_tail else []
    print(f"{k:46s} calls/decision {per:6.1f}   host ms/decision {sum(tail) / max(nd - half, 1) * 1e3:8.3f}")
