"""Helper of tests/test_nbv_gpu.py::test_sharded_step_two_ranks_matches_single_rank (launched with torch.distributed.run, 2 ranks)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from conftest import golden  # noqa: E402
import weights  # noqa: E402


def main():
    rank, lr = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    dist.init_process_group("nccl", device_id=dev)
    from macarons_amd.networks import SconeVis, SconeOcc
    from macarons_amd.nbv import nbv_step, ViewStateGrid
    occ, vis = SconeOcc(), SconeVis()
    sdo = weights.make_state_dict(weights.shapes_of(occ), 2)
    sdv = weights.make_state_dict(weights.shapes_of(vis), 1)
    sdo["linear3.bias"] = sdo["linear3.bias"] + np.float32(0.5)
    occ.load_state_dict({k: torch.from_numpy(v) for k, v in sdo.items()})
    vis.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()})
    occ, vis = occ.to(dev).eval(), vis.to(dev).eval()
    g = golden("e2e_grid_config2")
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    perms = [torch.from_numpy(g[f"perm{i}"].astype(np.int64)) for i in range(3)]
    r = nbv_step(occ, vis, T(g["pc"]), T(g["X"]), T(g["X_view"]), T(g["X_cam"]), ViewStateGrid(dev), occ_perms=perms, samples=T(g["samples"]))
    ok = int(r["nbv_idx"]) == int(g["nbv_idx"])
    ok = ok and float(np.abs(r["occ"].cpu().numpy() - g["occ"]).max()) < 1e-4 * float(np.abs(g["occ"]).max())
    c0, c1 = r["cam_range"]
    ok = ok and float(np.abs(r["gains"].cpu().numpy() - g["gains"][c0:c1]).max()) < 1e-4 * float(np.abs(g["gains"]).max())
    t = torch.tensor([int(ok)], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0 and int(t) == 1:
        print("TWO_RANK_OK")
    dist.destroy_process_group()
    sys.exit(0 if int(t) == 1 else 1)


if __name__ == "__main__":
    main()
