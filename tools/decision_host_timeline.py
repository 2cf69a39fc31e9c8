"""Dev: host-side timeline of ONE MACARONS decision of the bench scene (no extra synchronisations): when the host enters and leaves
every wrapped call, relative to the decision's start -- what the host thread does while the GPU's front section waits for it."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd.utility import macarons_utils as mu, scene as sc, scone_utils as su
from macarons_amd.networks import SconeOcc as _Occ
from macarons_amd import ops as _ops
log, depth = [], [0]
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); depth[0] += 1
        try:
            return f(*a, **k)
        finally:
            depth[0] -= 1; log.append((t0, time.perf_counter(), depth[0], name))
    setattr(mod, name, g)
for m, n in ((mu, "macarons_nbv_decision"), (mu, "compute_scene_occupancy_probability_field"), (mu, "predict_coverage_gain_for_cameras"),
             (mu, "_field_select"), (mu, "_field_prepare"), (mu, "_grid_tables"), (mu, "_store_of"), (su, "view_space_bin_permutation"),
             (sc.Scene, "fill_cells_begin"), (sc.Scene, "fill_counts"), (sc.Scene, "fill_cells_draw"), (sc.Scene, "fill_cells_apply"),
             (sc.Scene, "update_from_depth"), (sc.Scene, "set_all_features_to_value"), (_Occ, "forward_ragged_begin"), (_Occ, "forward_ragged_finish"),
             (_ops, "scone_occ_forward_ragged"), (_ops, "scone_vis_forward"), (_ops, "sample_proxy_batched"), (_ops, "points_in_fov"),
             (_ops, "field_build"), (_ops, "field_finish"), (_ops, "field_select"), (_ops, "scene_fill_begin"), (_ops, "h2d"),
             (_ops, "uniform_rows"), (_ops, "camera_boxes"), (_ops, "best_record"), (_ops, "proxy_scene_update_")):
    if hasattr(m, n):
        wrap(m, n)
os.environ["MCR_BENCH_NO_CHECKS"] = "1"
r = bench.measure_macarons_step(torch.device("cuda:0"))
print("p50 ms", r["p50_ms"])
dec = [e for e in log if e[3] == "macarons_nbv_decision"]
d0, d1 = dec[len(dec) // 2][0], dec[len(dec) // 2][1]
print(f"decision host span {1e3 * (d1 - d0):.3f} ms")
for t0, t1, dp, name in sorted(e for e in log if d0 <= e[0] and e[1] <= d1 and e[3] != "macarons_nbv_decision"):
    print(f"{1e6 * (t0 - d0):9.1f} us  +{1e6 * (t1 - t0):8.1f} us  {'  ' * dp}{name}")
