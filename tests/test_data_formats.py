"""Readers of the reference's on-disk formats + ground-truth coverage metrics (CPU; synthetic files with the
reference's schemas — real ShapeNet data is not in the container)."""
import json
import os

import numpy as np
import torch

from macarons_amd.utility import data as D


def test_shapenet_tensor_files_and_coverage_metrics(tmp_path):
    obj = tmp_path / "03001627" / "abc"
    (obj / "tensors").mkdir(parents=True)
    mesh_path = str(obj / "model.obj")
    g = torch.Generator().manual_seed(0)
    n_cam, n_surf = 6, 50
    cov = [(torch.rand(n_surf, generator=g) > 0.6).float() for _ in range(n_cam)]
    pcs = [torch.rand(10 + i, 3, generator=g) for i in range(n_cam)]
    torch.save({'partial_point_cloud': pcs, 'coverage': cov}, obj / "tensors" / "partial_point_clouds.pt")
    torch.save({'occupancy_field': torch.rand(20, 4, generator=g)}, obj / "tensors" / "occupancy_field.pt")
    torch.save({'surface_points': torch.rand(30, 3, generator=g), 'epsilon': 0.01}, obj / "tensors" / "surface_points.pt")
    part_pc, coverage = D.get_gt_partial_point_clouds(mesh_path, "cpu", normalization_factor=2.0)
    assert len(part_pc) == n_cam and coverage.shape == (n_cam, n_surf) and torch.allclose(part_pc[1], 2 * pcs[1])
    X, occ = D.get_gt_occupancy_field(mesh_path, "cpu")
    assert X.shape == (20, 3) and occ.shape == (20, 1)

    class P: surface_epsilon_is_constant = False; surface_epsilon = 0.5
    surf, eps = D.get_gt_surface(P, mesh_path, "cpu", normalization_factor=2.0)
    assert surf.shape == (30, 3) and abs(eps - 0.02) < 1e-9
    # coverage metrics vs a direct restatement with torch.heaviside (scone_utils.py:649-680)
    idx = torch.tensor([0, 3])
    prev = coverage[idx].sum(0)
    ref_cov = torch.heaviside(prev, torch.zeros_like(prev)).mean()
    assert torch.allclose(D.compute_surface_coverage_from_cam_idx(coverage, idx), ref_cov.view(1))
    gain = D.compute_gt_coverage_gain_from_precomputed_matrices(coverage, idx)
    ref = torch.stack([torch.heaviside(prev + coverage[c], torch.zeros_like(prev)).mean() - ref_cov for c in range(n_cam)])
    assert gain.shape == (n_cam, 1) and torch.allclose(gain[:, 0], ref)
    seqs = {"abc": {'idx': [3, 1, 4], 'coverage': [torch.tensor(0.1), torch.tensor(0.2), torch.tensor(0.3)]}}
    s, c = D.get_optimal_sequence(seqs, mesh_path, 2)
    assert s.tolist() == [3, 1] and len(c) == 2


def test_checkpoints_and_scene(tmp_path):
    from macarons_amd.networks import SconeVis, SconeOcc
    vis, occ = SconeVis(), SconeOcc()
    sd = {"module." + k: v + 1 for k, v in vis.state_dict().items()}
    torch.save({'epoch': 3, 'model_state_dict': sd, 'optimizer_state_dict': {}, 'loss': 0.1}, tmp_path / "vis.pth")
    v2 = D.load_weights(SconeVis(), str(tmp_path / "vis.pth"), ddp_model=True, device="cpu")
    assert torch.equal(v2.fc3.bias, vis.fc3.bias + 1)
    scone = {**{"occupancy." + k: v for k, v in occ.state_dict().items()}, **{"visibility." + k: v for k, v in vis.state_dict().items()}}
    torch.save({'model_state_dict': {'depth': {}, 'scone': scone}}, tmp_path / "macarons.pth")
    o3, v3 = D.load_scone_from_macarons_checkpoint(SconeOcc(), SconeVis(), str(tmp_path / "macarons.pth"), "cpu")
    assert torch.equal(o3.linear1.weight, occ.linear1.weight) and torch.equal(v3.fc1.weight, vis.fc1.weight)
    scene = tmp_path / "liberty"
    scene.mkdir()
    json.dump({"grid": {"l": 5}}, open(scene / "settings.json", "w"))
    torch.save({'X_idx': torch.zeros(4, 3), 'occupied': torch.ones(4)}, scene / "occupied_pose.pt")
    st, xi, oc = D.load_scene(str(scene))
    assert st["grid"]["l"] == 5 and xi.shape == (4, 3) and oc.shape == (4,)


def test_reference_shipped_files():
    """The files the reference itself ships (tests/golden/ref_data: a 6-object slice of
    data/ShapeNetCore.v1/validation_optimal_trajectories.pt in its own pickle schema, and data/scenes/liberty verbatim) read
    through macarons_amd.utility.data with weights_only=True give what the REFERENCE's loaders returned for them
    (expected.json, written by tests/golden/make_golden.py: gen_formats from scone_utils.py:639-646,699-711 and
    CustomDataset.py:313-362)."""
    root = os.path.join(os.path.dirname(__file__), "golden", "ref_data")
    exp = json.load(open(os.path.join(root, "expected.json")))
    seqs = D.get_validation_optimal_sequences(os.path.join(root, "validation_optimal_trajectories_slice.pt"))
    assert sorted(seqs.keys()) == sorted(exp["keys"]) and exp["n_objects_in_full_file"] == 399
    for k, e in exp["expected"].items():
        assert len(seqs[k]["idx"]) == 10 and len(seqs[k]["coverage"]) == 10
        idx, cov = D.get_optimal_sequence(seqs, f"/any/where/03001627/{k}/model.obj", 4)
        assert idx.dtype == torch.int64 and idx.tolist() == e["idx"] and [float(c) for c in cov] == e["coverage"]
    item = D.scene_item(os.path.join(root, "scenes"), "liberty")
    s = exp["scene"]
    assert item["scene_name"] == s["scene_name"] and item["obj_name"] == s["obj_name"] and item["settings"] == s["settings"]
    pose = item["occupied_pose"]
    assert list(pose["X_idx"].shape) == s["X_idx_shape"] and list(pose["occupied"].shape) == s["occupied_shape"]
    assert str(pose["X_idx"].dtype) == s["X_idx_dtype"] and str(pose["occupied"].dtype) == s["occupied_dtype"]
    assert int(pose["occupied"].sum()) == s["n_occupied"] and pose["X_idx"][:5].tolist() == s["X_idx_first"]
    assert pose["X_idx"][-1].tolist() == s["X_idx_last"]
    st, xi, oc = D.load_scene(os.path.join(root, "scenes", "liberty"))
    assert st == s["settings"] and torch.equal(xi, pose["X_idx"]) and torch.equal(oc, pose["occupied"])
    # the pose lattice of settings.json matches the occupied-pose table: pose_l * pose_w * pose_h entries
    cam = st["camera"]
    assert xi.shape[0] == cam["pose_l"] * cam["pose_w"] * cam["pose_h"]
