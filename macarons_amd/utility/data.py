"""On-disk formats of the reference, so real ShapeNet / scene data and checkpoints can be fed to
the MI355X path when present.  Host-side IO and ground-truth metrics only (plain torch; not on the hot path).

Mirrors (upstream tree):
  macarons/utility/scone_utils.py:571-593  get_gt_partial_point_clouds   tensors/partial_point_clouds.pt
        {'partial_point_cloud': list[n_cam] of [P_i,3], 'coverage': list[n_cam] of [n_surface]}
  macarons/utility/scone_utils.py:596-613  get_gt_occupancy_field        tensors/occupancy_field.pt {'occupancy_field': [K,4]}
  macarons/utility/scone_utils.py:616-633  get_gt_surface                tensors/surface_points.pt {'surface_points','epsilon'}
  macarons/utility/scone_utils.py:636-646  get_optimal_sequence          validation_optimal_trajectories.pt
  macarons/utility/scone_utils.py:649-680  compute_gt_coverage_gain_from_precomputed_matrices, compute_surface_coverage_from_cam_idx
  macarons/utility/utils.py:140-185        load_ddp_state_dict / load_weights  (checkpoint dicts, optional 'module.' prefix)
  macarons/networks/Macarons.py:38-52,85-104  nested {'depth':…, 'scone':…} checkpoints with 'occupancy.' / 'visibility.' keys
  macarons/utility/CustomDataset.py:313-362   scene dirs: settings.json + occupied_pose.pt {'X_idx': [n,3], 'occupied': [n]}
"""
import json
import os
import re
from collections import OrderedDict

import torch


def _load(path, device, trusted=False):
    """torch.load restricted to tensors / plain containers (weights_only=True): every file of the reference's formats is a
    dict / list of tensors and scalars.  trusted=True unpickles arbitrary objects (only for files you produced yourself)."""
    return torch.load(path, map_location=device, weights_only=not trusted)


def _tensors_dir(path):
    return os.path.join(os.path.dirname(path), "tensors")


def get_gt_partial_point_clouds(path, device, normalization_factor=None):
    pc_dict = _load(os.path.join(_tensors_dir(path), "partial_point_clouds.pt"), device)
    part_pc = pc_dict['partial_point_cloud']
    coverage = torch.vstack(pc_dict['coverage'])
    if (normalization_factor is not None) and (normalization_factor != 1.):
        for i in range(len(part_pc)):
            part_pc[i] = normalization_factor * part_pc[i]
    return part_pc, coverage


def get_gt_occupancy_field(path, device):
    pc_dict = _load(os.path.join(_tensors_dir(path), "occupancy_field.pt"), device)
    return pc_dict['occupancy_field'][..., :3], pc_dict['occupancy_field'][..., 3:]


def get_gt_surface(params, path, device, normalization_factor=None):
    d = _load(os.path.join(_tensors_dir(path), "surface_points.pt"), device)
    gt_surface = d['surface_points']
    eps = params.surface_epsilon if params.surface_epsilon_is_constant else d['epsilon']
    if (normalization_factor is not None) and (normalization_factor != 1.):
        gt_surface, eps = gt_surface * normalization_factor, eps * normalization_factor
    return gt_surface, eps


def get_validation_optimal_sequences(path, device="cpu"):
    """scone_utils.py:699-711: the dict {object id: {'idx': [10 camera ids], 'coverage': [10 tensors]}} of greedy-optimal view
    sequences (data/ShapeNetCore.v1/validation_optimal_trajectories.pt in the reference tree; `path` = that file)."""
    return _load(path, device)


def get_optimal_sequence(optimal_sequences, mesh_path, n_views):
    key = os.path.basename(os.path.dirname(mesh_path))
    optimal_seq = torch.Tensor(optimal_sequences[key]['idx']).long()
    return optimal_seq[:n_views], optimal_sequences[key]['coverage'][:n_views]


def compute_gt_coverage_gain_from_precomputed_matrices(coverage, initial_cam_idx):
    """coverage [n_cam, n_surface] 0/1 -> gain [n_cam, 1] of adding each camera to the initial set."""
    n_cam, n_pts = coverage.shape
    prev = torch.sum(coverage[initial_cam_idx], dim=0).view(1, n_pts)
    previous_coverage = torch.mean((prev > 0).to(coverage.dtype), dim=-1)
    new = torch.mean(((prev + coverage) > 0).to(coverage.dtype), dim=-1)
    return (new - previous_coverage).view(-1, 1)


def compute_surface_coverage_from_cam_idx(coverage, cam_idx):
    return torch.mean((torch.sum(coverage[cam_idx], dim=0) > 0).to(coverage.dtype), dim=-1).view(1)


def strip_ddp_prefix(state_dict):
    """utils.py:140-158: drop a leading 'module.' from every key (checkpoints written from DDP-wrapped models)."""
    if not any(re.search("module", k) for k in state_dict):
        return state_dict
    pattern = re.compile('module.')
    return OrderedDict((re.sub(pattern, '', k), v) for k, v in state_dict.items())


def load_weights(model, trained_weights_file, ddp_model, device):
    """utils.py:161-185: checkpoint dict {'epoch','model_state_dict','optimizer_state_dict','loss',...}."""
    model = model.to(device)
    checkpoint = _load(trained_weights_file, device)
    sd = checkpoint['model_state_dict']
    model.load_state_dict(strip_ddp_prefix(sd) if ddp_model else sd)
    return model


def load_scone_from_macarons_checkpoint(occupancy_model, visibility_model, checkpoint_file, device):
    """pretrained_macarons.pth-style checkpoints nest {'depth': …, 'scone': {'occupancy.*', 'visibility.*'}}
    (Macarons.py:38-52, 85-104).  Loads the two SCONE modules, ignores the depth net (out of scope)."""
    ck = _load(checkpoint_file, device)
    sd = ck['model_state_dict'] if 'model_state_dict' in ck else ck
    scone = strip_ddp_prefix(sd['scone'] if 'scone' in sd else sd)
    occupancy_model.load_state_dict(OrderedDict((k[len('occupancy.'):], v) for k, v in scone.items() if k.startswith('occupancy.')))
    visibility_model.load_state_dict(OrderedDict((k[len('visibility.'):], v) for k, v in scone.items() if k.startswith('visibility.')))
    return occupancy_model.to(device), visibility_model.to(device)


def load_scene(scene_dir, device="cpu"):
    """CustomDataset.py:313-362: a scene directory holds settings.json and occupied_pose.pt."""
    with open(os.path.join(scene_dir, "settings.json")) as f:
        settings = json.load(f)
    pose = _load(os.path.join(scene_dir, "occupied_pose.pt"), device)
    return settings, pose['X_idx'], pose['occupied']


def scene_item(data_path, scene_name, use_occupied_pose=True):
    """SceneDataset.__getitem__ (CustomDataset.py:337-362): {'scene_name', 'obj_name', 'settings'[, 'occupied_pose']}; the mesh
    name is the first *.obj in the directory, '<scene>.obj' if there is none."""
    scene_path = os.path.join(data_path, scene_name)
    obj_name = scene_name + '.obj'
    for file_name in os.listdir(scene_path):
        if file_name[-4:] == '.obj':
            obj_name = file_name
            break
    with open(os.path.join(scene_path, 'settings.json'), "r") as f:
        settings = json.load(f)
    scene = {'scene_name': scene_name, 'obj_name': obj_name, 'settings': settings}
    if use_occupied_pose:
        scene['occupied_pose'] = _load(os.path.join(scene_path, 'occupied_pose.pt'), "cpu")
    return scene
