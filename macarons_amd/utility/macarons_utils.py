"""Mirror of the MACARONS-regime scoring helpers of macarons/utility/macarons_utils.py on the MI355X kernels:
  get_distance_factor :1741-1765, get_distance_factor_threshold :1768-1776, get_distance_factor_smooth :1779-1788
  predict_coverage_gain_for_single_camera :1580-1738  (here batched over the <= 30 neighbour cameras)
PyTorch3D camera objects stay outside: cameras are passed as the 40-float records of mcr_points_in_fov
(M_view, M_proj, ndc bounds, centre, range) and a prediction-view matrix.
"""
import os

import numpy as np
import torch

from .. import ops
from .scone_utils import sample_proxy_points  # noqa: F401  (same sampler as the SCONE regime)


def camera_record(M_view, M_proj, ndc_bounds, center, fov_range=None):
    """Pack one camera for points_in_fov: 4x4 row-vector matrices (pytorch3d get_matrix()[0] layout)."""
    rec = torch.zeros(40, dtype=torch.float32)
    rec[:16] = torch.as_tensor(M_view, dtype=torch.float32).reshape(-1)
    rec[16:32] = torch.as_tensor(M_proj, dtype=torch.float32).reshape(-1)
    rec[32:36] = torch.as_tensor(ndc_bounds, dtype=torch.float32)
    rec[36:39] = torch.as_tensor(center, dtype=torch.float32)
    rec[39] = 0.0 if fov_range is None else float(fov_range)
    return rec


def _factor(pts, X_cam, distance_th, smooth):
    n = pts.shape[0]
    ones = torch.ones(1, n, dtype=torch.float32, device=pts.device)
    ops.macarons_gain_(ones, pts.reshape(1, n, -1)[..., :3].contiguous(), X_cam.reshape(1, 3).contiguous(),
                       torch.ones(1, dtype=torch.float32, device=pts.device), distance_th, smooth)
    return ones.view(n, 1)


def get_distance_factor_threshold(pts, X_cam, distance_th=17.):
    """[n_pts, 1] factor min(1, (th/d)^2)  (macarons_utils.py:1768-1776)."""
    return _factor(pts, X_cam, distance_th, False)


def sensor_distance_threshold(params, fov_camera, cell_resolution):
    """distance_th = focal_length * epsilon / pixel_size of macarons_utils.py:1752-1755 (= :1780-1783)."""
    import math
    fov = float(torch.as_tensor(fov_camera.fov).reshape(-1)[0])
    focal_length = 1. / math.tan(math.pi / 180. * fov / 2.)
    pixel_size = 2. / min(params.image_height, params.image_width)
    epsilon = math.sqrt(math.pi) / 2. * cell_resolution
    return focal_length * epsilon / pixel_size


def get_distance_factor(params, pts, X_cam, fov_camera, cell_resolution):
    """macarons_utils.py:1741-1765: 1 within distance_th, epsilon^2 (focal / pixel / d)^2 = (distance_th / d)^2 beyond."""
    return _factor(pts, X_cam, sensor_distance_threshold(params, fov_camera, cell_resolution), False)


def get_distance_factor_smooth(params, pts, X_cam, fov_camera, cell_resolution):
    """macarons_utils.py:1779-1788: 1 / (1 + (d / distance_th)^2)."""
    return _factor(pts, X_cam, sensor_distance_threshold(params, fov_camera, cell_resolution), True)


def predict_coverage_gain_for_cameras(visibility_model, X_world, proxy_view_harmonics, occ_probs, cameras, X_cam_world,
                                      prediction_view_matrices, prediction_box_diag, seq_len=2048, min_occ=0.1,
                                      distance_th=17., samples=None, smooth=False, return_parts=False, record=None,
                                      uniform_draws="per_camera"):
    """The per-neighbour-camera scoring loop of testers/scene.py:434-454 around
    predict_coverage_gain_for_single_camera (macarons_utils.py:1580-1738), for K cameras AT ONCE and without a host
    synchronisation (the reference runs one SconeVis forward and reads a count back per camera):
      frustum mask (all K at once) -> occupancy-weighted sampling inside each frustum -> prediction-view space ->
      SconeVis -> per-point visibility gains (C = 1) x distance factor -> mean x sum(occ in frustum).
    X_world [P,3], proxy_view_harmonics [P,64], occ_probs [P,1], cameras [K,40], X_cam_world [K,3],
    prediction_view_matrices [K,4,4] (world -> prediction-camera view, row-vector).  distance_th / smooth select the distance
    factor (params.distance_factor_th: a number -> threshold factor; None -> sensor_distance_threshold(...), smooth=False;
    'smooth' -> sensor_distance_threshold(...), smooth=True).  Returns gains [K] (and, with return_parts, the per-camera lists of
    factored per-point gains [N] and sampled world points [N,4]).
    Uniforms (`samples` None): uniform_draws="per_camera" (default) = what upstream's K calls draw one after the other,
    torch.rand(S, 1, device=...) inside every per-camera call (scone_utils.py:1052), bit for bit and with the same effect on the device
    generator -- from ONE launch (ops.uniform_rows: torch's Philox indexing restated); "batched" = one torch.rand(K, S) (another stream)."""
    K = cameras.shape[0]
    dev = X_world.device
    S = seq_len
    mask = ops.points_in_fov(X_world, cameras)                                            # :1603
    occ_k = ops.fov_mask_occ(mask, occ_probs.reshape(-1).contiguous())                    # :1606-1613 folded into the sampler
    # ---- sampling inside every frustum (:1624): K distributions over the ONE shared point set in one launch sequence, nothing
    # read back (padded rows, counts on the device)
    if samples is None:
        u = ops.uniform_rows(K, S, dev) if uniform_draws == "per_camera" else torch.rand(K, S, device=dev)
    elif torch.is_tensor(samples):
        u = samples.to(dev).reshape(K, S).float()
    else:
        u = torch.stack([torch.as_tensor(x, device=dev).reshape(-1) for x in samples]).float()
    if record is not None:
        record["samples"] = u                                                               # (a re-run must see the same uniforms)
    res, res_h, inv, _, nu, vol = ops.sample_proxy_batched(X_world, occ_k, proxy_view_harmonics, u.contiguous(), min_occ)
    vol = vol.float()                                                                      # [K,S,4] [K,S,64] [K,S]; [K] int32, [K]
    # ---- prediction boxes (:1631-1641) and the cameras in the normalised prediction space (:1655-1659): one launch
    Mv = prediction_view_matrices.to(device=dev, dtype=torch.float32).contiguous()
    xc = X_cam_world.reshape(K, 3).contiguous()
    inv_d = 1.0 / prediction_box_diag
    center, cam_view = ops.camera_boxes(res, nu, Mv, xc, inv_d)
    pts = res.clone()
    ops.transform_points_batched_(pts, Mv, center, torch.full((K,), inv_d, dtype=torch.float32, device=dev))   # :1647-1650, all cameras in one launch
    # ---- ONE SconeVis forward over the K padded clouds (:1664), ONE scorer launch on the UNIQUE points (C = 1 per cloud, :1683), ONE
    # gain launch that reads the Monte-Carlo duplicates (:1668-1671) through the inverse map
    harm = visibility_model(pts, view_harmonics=res_h, lengths=nu)
    vis_u = ops.sh_visibilities(pts, harm, cam_view.view(K, 1, 3), True).view(K, S)
    if return_parts:
        gi = inv[..., None]
        world = torch.gather(res, 1, gi.expand(-1, -1, 4)).contiguous()
        vis = torch.gather(vis_u, 1, inv).contiguous()
        gains = ops.macarons_gain_(vis, world, xc, vol, distance_th, smooth)               # :1699-1704 (vis scaled in place)
        gains = torch.where(nu > 0, gains, torch.zeros_like(gains))                       # empty frustum: gain 0 (:1707-1736)
        n_host = nu.tolist()
        return (gains, [vis[k] if n_host[k] > 0 else None for k in range(K)],
                [world[k] if n_host[k] > 0 else None for k in range(K)])
    return ops.macarons_gain_indexed(vis_u, res, inv, nu, xc, vol, distance_th, smooth)   # :1699-1704; empty frustum: gain 0 (:1707-1736)


class SceneCamera:
    """What one MACARONS decision reads of the reference's Camera / PyTorch3D objects (which stay outside the kernels):
    the 40-float record of mcr_points_in_fov (world->view matrix, full projection matrix, NDC bounds, centre, sensor range), the
    camera centre X_cam [1,3], zfar and the field of view in degrees."""

    def __init__(self, record, X_cam, zfar, fov=60.0):
        self.record, self.X_cam, self.zfar = record.reshape(40).contiguous(), X_cam.reshape(1, 3).contiguous(), float(zfar)
        self.fov = torch.tensor([float(fov)])
        self._m_view_host = self.record[:16].view(4, 4) if self.record.device.type == "cpu" else None

    @property
    def M_view(self):
        return self.record[:16].view(4, 4)

    @property
    def M_view_host(self):
        """The world->view matrix on the host: the record's own memory when the caller keeps the record on the host (a pose is host
        data: hand the record over as a CPU tensor and the decision's host-side geometry never waits for the GPU), else ONE read-back
        per camera object, kept."""
        if self._m_view_host is None:
            self._m_view_host = self.record[:16].view(4, 4).cpu()
        return self._m_view_host


def macarons_nbv_decision(params, macarons, proxy_scene, surface_scene, camera, depth, depth_mask, neighbor_records, X_neighbors,
                          device, samples=None, return_signed_distances=False, range_guard=True, group=None, uniform_draws="per_camera"):
    """One next-best-view decision of the MACARONS loop after the depth map of the current pose is known -- the body of
    testers/scene.py:391-454 (everything between the depth network and the move to the chosen pose):
      1. proxy points in the current frustum (Camera.get_points_in_fov :391), registered in the proxy grid (:394-395);
      2. signed distances to the depth map, view-state OR, supervision occupancy, out-of-field flags (:397-415) -- one fused pass;
      3. surface features reset (:418); occupancy-probability field over the seen cells (:421-425);
      4. coverage gain of every valid neighbour pose (:434-450), all cameras in one launch sequence; first strict maximum (:452-454).
    camera: SceneCamera of the current pose (it is also the prediction camera, fov_camera_0 of :305); depth [H,W] (+ optional
    leading/trailing singleton dims), depth_mask like depth; neighbor_records [K,40], X_neighbors [K,3].
    `group` (torch.distributed; every rank holds replicas of both scenes and calls with the same arguments; None = a local decision
    whatever process groups exist): SURVEY §8e -- the query rows of the occupancy field and the K neighbour cameras are
    block-partitioned over the ranks, the occupancies (4 B per proxy point) and one 8-byte (gain, index) record per rank are
    all-gathered, the hidden draws (Cell.fill subsets, SconeOcc's down-samples, the sampling uniforms) are rank 0's; the cheap state
    updates run replicated.  Bit for bit the 1-rank decision; `gains` then holds this rank's cameras only (`cam_range`).
    The hidden permutations (Cell.fill's subsets, SconeOcc's down-samples) are drawn with torch.randperm on the CPU generator in
    upstream's order -- what the reference goldens pin (an opt-in device-generator source existed until round 6: slower than this path
    once the draws had become two C++ calls, and removed).  uniform_draws: see predict_coverage_gain_for_cameras.
    Returns dict(next_idx (device int64: index into the neighbour list), gains [K], fov_mask [P] bool, X_world, view_harmonics,
    occ_probs).  The scene objects are updated in place like upstream.
    Host synchronisations: ONE in the middle (the per-cell counts of fill_cells and of the field's selection come back together: the
    shapes of everything behind them depend on them) and -- range_guard=True -- one at the end (the range flag of the fp16-split path
    with the decision's record; range_guard=False: the caller checks macarons.occupancy.range_flag() itself).  The reference's order of
    CPU-generator draws is kept (Cell.fill's, then SconeOcc's job by job) while the first launches of the occupancy pass, which need
    none of them, are already queued."""
    from .. import dist as mdist
    world, rank = mdist.group_world_rank(group)            # group=None: local, whatever process groups exist
    xch = mdist.exchange_on(group)                         # the exchange path (rank 0's draws, all-gathers, record merge) runs
    H, W = params.image_height, params.image_width
    depth2 = depth.reshape(H, W).contiguous().float()
    dmask2 = depth_mask.reshape(H, W) if depth_mask is not None else None
    ps = proxy_scene
    # the field pass's host geometry (every cell's prediction transform, the bin permutation) starts NOW on the extension's worker thread
    Mv_field = camera.M_view_host if hasattr(camera, "M_view_host") else camera.M_view
    prep_ticket = _field_prepare_begin(params, ps, Mv_field, device) if hasattr(ps, "fill_cells_begin") else None
    rec = ops.h2d(camera.record, torch.float32, device)
    # 1 ---- proxy points in the current field of view, offered to their grid cells with their index as feature (every point is offered,
    # the ones outside the frustum flagged invalid: compacting them first -- `points[mask]` -- is a read-back of the count)
    fov_mask = ops.points_in_fov(ps.proxy_points, rec.view(1, 40))[0]
    P_all = ps.proxy_points.shape[0]
    idx_f = getattr(ps, "_mcr_index_feature", None)
    if idx_f is None or idx_f.shape[0] != P_all or idx_f.device != ps.proxy_points.device:
        idx_f = torch.arange(P_all, device=device, dtype=torch.float32).view(-1, 1)
        try:
            ps._mcr_index_feature = idx_f
        except Exception:
            pass
    fused = hasattr(ps, "fill_cells_begin")
    if fused:
        fill = ps.fill_cells_begin(ps.proxy_points, features=idx_f, valid=fov_mask)
    else:                                                   # a reference Scene: its own fill (one Python loop over the cells)
        ps.fill_cells(ps.proxy_points[fov_mask], features=idx_f[fov_mask])
    # 2 ---- carve with the depth map: signed distance, view states, supervision occupancy, out-of-field, one launch
    x_cam = rec[36:39].view(1, 3) if camera.X_cam.device.type == "cpu" and torch.equal(camera.X_cam.reshape(3), camera.record[36:39]) \
        else ops.h2d(camera.X_cam, torch.float32, device)    # (a host record carries the centre: one upload instead of two)
    sgn = ps.update_from_depth(fov_mask, rec, x_cam, depth2, dmask2, fill=1.1 * camera.zfar,
                               tol=params.carving_tolerance, return_signed_distances=return_signed_distances)
    surface_scene.set_all_features_to_value(value=1.)
    # 3 ---- occupancy probability field, in the current camera's view space; 4 ---- neighbours.
    # The range check of the fp16-split path (SconeOcc.range_guard) is DEFERRED to one read-back at the very end: checked inside
    # the occupancy pass it stalls the host until the pass has run.  On a set flag steps 3-4 are repeated on the full-range
    # variant with the SAME hidden draws (cell permutations, sampling uniforms).
    # the world->view matrix where the host-side geometry needs it: a record kept on the host costs nothing, a device record ONE
    # read-back per camera object (a read-back per decision stalled the host behind the fill / selection launches: 0.2-0.3 ms)
    Mv = rec[:16].view(4, 4)                          # (the device copy of the record already holds it: no second upload)
    K = neighbor_records.shape[0]
    th = params.distance_factor_th
    smooth = th == 'smooth'
    if th is None or smooth:
        th = sensor_distance_threshold(params, camera, surface_scene.cell_resolution)
    vis_model = macarons.visibility                 # `macarons` = the SCONE part (Macarons.scone upstream): .occupancy / .visibility
    diag = getattr(ps, "_mcr_box_diag", None)       # a constant of the scene: read back once, not once per decision
    if diag is None:
        diag = torch.linalg.norm(ps.x_max - ps.x_min).item()
        try:
            ps._mcr_box_diag = diag
        except Exception:
            pass
    occ_net = getattr(macarons, "occupancy", macarons)
    nrec, xn = ops.h2d(neighbor_records, torch.float32, device), ops.h2d(X_neighbors, torch.float32, device)
    k0, k1 = mdist.shard_range(K, rank, world)
    S = params.seq_len
    # ---- the selection of the field pass is queued BEHIND the fill's device part and BEFORE its host part: with no cell over its
    # capacity the set a cell ends up with does not depend on the draws (only its order does), so both count tables come back in ONE
    # read-back, and the fill's draws + gather (fill_cells_end) run while the GPU already works on the occupancy pass
    field_state = {"selection": None, "between": None, "after": None}
    if fused:
        sel = _field_select(ps, device, True, pending=fill)
        prep = _field_prepare(params, ps, Mv_field, device, prep_ticket)   # (count-independent host work: from the worker thread)
        nkf = 4 * fill.nk + 6
        host = torch.cat((fill.counts, sel.counts)).cpu().numpy()                                  # THE read-back of the decision's first half
        cand, adm = ps.fill_counts(host[:nkf])
        gfill = group if xch else None
        if ps.fill_overflows(cand, adm):            # a full cell: WHICH points stay is random -> the selection has to wait for the draws
            ps.fill_cells_end(fill, cand, adm, 0, gfill)
        else:
            plan = {}
            field_state["selection"] = (sel, host[nkf:], prep)
            # the fill's draws come before the occupancy pass's on the CPU generator (upstream's order); its gather can wait until the
            # pass is queued -- nothing of THIS decision reads the proxy cells any more
            def _draw_fill():
                if "p" not in plan:
                    plan["p"] = ps.fill_cells_draw(fill, cand, adm, 0, gfill)

            def _apply_fill():                          # (once: the plan is consumed)
                if not plan.get("applied"):
                    plan["applied"] = True
                    ps.fill_cells_apply(plan.get("p"))
            field_state["between"], field_state["after"] = _draw_fill, _apply_fill

    def field_and_gains(ragged_perms, smp, record):
        selection, between, after = field_state["selection"], field_state["between"], field_state["after"]
        field_state["selection"] = field_state["between"] = field_state["after"] = None   # (a repeat selects again: the stores are final by then)
        try:
            X_world, view_harmonics, occ_probs = compute_scene_occupancy_probability_field(params, macarons, None, surface_scene, ps,
                                                                                           device, prediction_camera=Mv_field, ragged_perms=ragged_perms,
                                                                                           group=group if xch else None, record=record,
                                                                                           _selection=selection,
                                                                                           _between=between, _after=after)
        except BaseException:
            # the pass raised between the fill's two halves (the draws may or may not have been made, the gather has not run): finish the
            # fill -- draw if that is still due, then apply -- so that the proxy scene has received this frustum's points as upstream's
            # Cell.fill would have left it before the occupancy pass started, and the CPU generator is where upstream's would be
            if between is not None:
                between()
            if after is not None:
                after()
            raise
        if xch:                                         # the uniforms of ALL cameras are rank 0's
            if smp is None:
                smp = ops.uniform_rows(K, S, device) if uniform_draws == "per_camera" else torch.rand(K, S, device=device)
                _, smp = mdist.broadcast_draws([], smp, 0, group)
            elif not torch.is_tensor(smp):
                smp = torch.stack([torch.as_tensor(x_, device=device).reshape(-1) for x_ in smp]).float()
            smp = smp.to(device).reshape(K, S).float()
            if record is not None:
                record["samples_all"] = smp
            if k1 > k0:
                gains = predict_coverage_gain_for_cameras(vis_model, X_world, view_harmonics, occ_probs, nrec[k0:k1].contiguous(), xn[k0:k1].contiguous(),
                                                          Mv.reshape(1, 4, 4).expand(k1 - k0, -1, -1), diag, seq_len=S,
                                                          min_occ=params.min_occ_for_proxy_points, distance_th=float(th), samples=smp[k0:k1].contiguous(),
                                                          smooth=smooth)
            else:                                       # empty camera shard (K < world)
                gains = torch.zeros(0, dtype=torch.float32, device=device)
            return X_world, view_harmonics, occ_probs, gains
        gains = predict_coverage_gain_for_cameras(vis_model, X_world, view_harmonics, occ_probs, nrec, xn,
                                                  Mv.reshape(1, 4, 4).expand(K, -1, -1), diag, seq_len=S,
                                                  min_occ=params.min_occ_for_proxy_points, distance_th=float(th), samples=smp,
                                                  smooth=smooth, record=record, uniform_draws=uniform_draws)
        if record is not None and "samples" in record:
            record["samples_all"] = record["samples"]
        return X_world, view_harmonics, occ_probs, gains

    guard_before = getattr(occ_net, "range_guard", None)
    deferred = range_guard and guard_before in ("sync", "async") and hasattr(occ_net, "forward_ragged")
    record = {}
    vis_prev = None
    if deferred:
        occ_net.range_guard = "defer"
        occ_net.clear_range_flag(device)                # (exists on every rank before the pass: the all-reduce below is rank-invariant)
        if hasattr(vis_model, "range_guard"):           # SconeVis reports into the SAME flag: one read-back covers both networks
            vis_prev = (vis_model.range_guard, vis_model._range_flag)
            vis_model.range_guard, vis_model._range_flag = "defer", occ_net.range_flag()
    try:
        X_world, view_harmonics, occ_probs, gains = field_and_gains(None, samples, record)
        # `if coverage_gain > max_coverage_gain` from -1: the first strict maximum (a NaN gain never wins upstream; here it would)
        rec_best = None if xch else ops.best_record(gains.view(1, K), 0)
        fallback = None
        if deferred:
            flag = occ_net.range_flag()
            if xch:
                flag = mdist.all_reduce_max(flag, group)                # every rank repeats, or none
            hit = False
            if flag is not None:                    # the one read-back of the second half: range flag + the decision's record together
                both = torch.cat((flag.view(1).float(), rec_best.view(-1))).cpu() if rec_best is not None else flag.cpu()
                hit = bool(both.view(-1)[0] != 0)
            if hit:                                 # out of the fp16 range: repeat on the full-range variant
                with ops.variant(5):                 # (scoped to this thread's calls: the process default is not touched)
                    X_world, view_harmonics, occ_probs, gains = field_and_gains(record.get("ragged_perms"), record.get("samples_all"), None)
                occ_net.clear_range_flag()
                fallback = 5
                rec_best = None if xch else ops.best_record(gains.view(1, K), 0)
    finally:
        if deferred:
            occ_net.range_guard = guard_before
        if vis_prev is not None:
            vis_model.range_guard, vis_model._range_flag = vis_prev
    if xch:                                             # ties -> the lowest index over all ranks = the first strict maximum
        max_gain, next_idx = mdist.allgather_best(gains.view(1, -1), k0, group)
        out = {"next_idx": next_idx[0], "max_gain": max_gain[0], "cam_range": (k0, k1)}
    else:
        out = {"next_idx": rec_best[0, 1].to(torch.int64), "max_gain": rec_best[0, 0]}
        if deferred and fallback is None and flag is not None:      # the decision came back with the range flag: no second read-back needed
            out["host"] = {"max_gain": float(both[1]), "next_idx": int(both[2])}
    out.update({"gains": gains, "fov_mask": fov_mask, "X_world": X_world, "view_harmonics": view_harmonics, "occ_probs": occ_probs})
    if fallback:
        out["fallback_variant"] = fallback
    if return_signed_distances:
        out["signed_distances"] = sgn              # [P], 0 outside the frustum
    return out


# ---- scene-side point bookkeeping -----------------------------------------------------------
def depth_camera_record(M_full_projection, k22, k32):
    """18 floats for ops.unproject_depth: inverse of the full (world->view->ndc) projection matrix + the two
    projection entries used by pytorch3d's scaled-depth conversion."""
    Minv = torch.linalg.inv(torch.as_tensor(M_full_projection, dtype=torch.float64)).float().reshape(-1)
    return torch.cat((Minv, torch.tensor([k22, k32], dtype=torch.float32)))


def compute_partial_point_cloud(depth, mask, camera, gathering_factor, fov_range=None, perm=None):
    """Camera.compute_partial_point_cloud (macarons_utils.py:2362-2398): depth [1,H,W,1] -> world points of the pixels
    kept by `mask` (and depth < fov_range), then a random `gathering_factor` fraction (torch.randperm on the CPU
    generator like the reference, or `perm`)."""
    H, W = depth.shape[1], depth.shape[2]
    pts = ops.unproject_depth(depth.reshape(1, H, W).contiguous(), camera.reshape(1, 18).to(depth.device))[0]
    keep = mask.reshape(-1).bool()
    if fov_range is not None:
        keep = keep & (depth.reshape(-1) < fov_range)
    world = pts[keep]
    n_points = int(len(world) * gathering_factor)
    idx = (torch.randperm(len(world)) if perm is None else perm)[:n_points]
    return world[idx.to(world.device)]


def project_depth_back_to_3D(depth, cameras):
    """utils.project_depth_back_to_3D (utils.py:1458-1487): depth [n_cam,H,W,1], cameras [n_cam,18] (depth_camera_record) ->
    world points of the pixels with depth > -1, camera-major."""
    n, H, W = depth.shape[0], depth.shape[1], depth.shape[2]
    pts = ops.unproject_depth(depth.reshape(n, H, W).contiguous(), cameras.to(depth.device))
    return pts[(depth > -1).view(n, -1)]


def cell_fill(cell_pts, pts, x_min, x_max, resolution, capacity, n_point_min=0, perm=None):
    """Cell.fill (macarons_utils.py:2551-2577) as a function of the cell state: strict bounding-box masks, fp64 admission test
    against the points already in the cell (HIP kernel), append, then keep a random `capacity` subset (torch.randperm on the
    CPU generator like the reference, or `perm`).  Returns the new cell points."""
    mask = torch.max(pts - x_max.view(1, 3), dim=-1)[0] < 0.
    add = pts[mask]
    if add.shape[0] == 0:
        return cell_pts
    add = add[torch.min(add - x_min.view(1, 3), dim=-1)[0] > 0.]
    if add.shape[0] <= n_point_min:
        return cell_pts
    if cell_pts.shape[0] > 0:
        add = add[cell_fill_mask(add.contiguous(), cell_pts.contiguous(), resolution)]
    out = torch.vstack((cell_pts, add))
    idx = (torch.randperm(len(out)) if perm is None else perm)[:capacity]
    return out[idx.to(out.device)]


def cell_fill_mask(pts_to_add, cell_pts, resolution, a_offsets=None, b_offsets=None):
    """The admission test of Cell.fill (macarons_utils.py:2565-2568): keep a candidate iff its fp64 distance to every
    point already in the (same) cell exceeds `resolution`.  With offsets, many cells are tested in one launch."""
    dev = pts_to_add.device
    if a_offsets is None:
        a_offsets = torch.tensor([0, pts_to_add.shape[0]], dtype=torch.int64, device=dev)
        b_offsets = torch.tensor([0, cell_pts.shape[0]], dtype=torch.int64, device=dev)
    return ops.min_dist_segmented(pts_to_add, a_offsets, cell_pts, b_offsets) > resolution


def covered_mask(surface_pts, seen_pts, epsilon, a_offsets=None, b_offsets=None, fp32_compare=False):
    """heaviside(epsilon - min cdist, 0) of camera_coverage_gain / scene_coverage (macarons_utils.py:3022-3024,
    3049-3051): a surface point is covered iff some seen point lies strictly within epsilon.  The nearest distance is fp64;
    scene_coverage compares it in fp64, camera_coverage_gain rounds it to fp32 first (`.float()`, :3022) and subtracts it from
    epsilon in fp32 (fp32_compare=True)."""
    dev = surface_pts.device
    if a_offsets is None:
        a_offsets = torch.tensor([0, surface_pts.shape[0]], dtype=torch.int64, device=dev)
        b_offsets = torch.tensor([0, seen_pts.shape[0]], dtype=torch.int64, device=dev)
    d = ops.min_dist_segmented(surface_pts, a_offsets, seen_pts, b_offsets)
    if fp32_compare:
        return (epsilon - d.float()) > 0.
    return (epsilon - d) > 0.


# ---- occupancy field of a scene: the per-cell SconeOcc pass of the MACARONS loop ----------------------------------
def _world_to_view_matrix(prediction_camera):
    """[4,4] row-vector world -> view matrix (X_view = [x y z 1] M) of a PyTorch3D camera object (its
    get_world_to_view_transform().get_matrix()) or the matrix itself."""
    if torch.is_tensor(prediction_camera):
        return prediction_camera.reshape(4, 4)
    return prediction_camera.get_world_to_view_transform().get_matrix().reshape(-1, 4, 4)[0]


def _lin(idx3, gw, gh):
    return (idx3[..., 0] * gw + idx3[..., 1]) * gh + idx3[..., 2]


def _grid_tables(scene, device):
    """What never changes for a scene grid, built once and kept on the scene object: the cell keys in linear-id (= lexicographic)
    order, key -> linear id, every cell's 27-neighbourhood (clamped, unique, sorted: get_neighboring_cells), and device tables of
    the cell centres and box diagonals."""
    tab = getattr(scene, "_mcr_grid_tables", None)
    if tab is not None and tab["device"] == str(device):
        return tab
    gl, gw, gh = scene.grid_l, scene.grid_w, scene.grid_h
    parse = lambda k: tuple(int(v) for v in k.strip("[]").split(","))
    keys = sorted(scene.cells.keys(), key=parse)
    lin_of = {k: (parse(k)[0] * gw + parse(k)[1]) * gh + parse(k)[2] for k in keys}
    by_lin = {lin_of[k]: scene.cells[k] for k in keys}
    n_cells = gl * gw * gh

    def neigh(c):
        i, j, k = c // (gw * gh), (c // gh) % gw, c % gh
        return sorted({(min(max(i + a, 0), gl - 1) * gw + min(max(j + b, 0), gw - 1)) * gh + min(max(k + d, 0), gh - 1)
                       for a in (-1, 0, 1) for b in (-1, 0, 1) for d in (-1, 0, 1)})
    order = [by_lin[c] for c in range(n_cells)]
    centers = torch.stack([c.center.reshape(3) for c in order]).to(device)
    diag = torch.linalg.norm(torch.stack([c.x_max.reshape(3) for c in order]) - torch.stack([c.x_min.reshape(3) for c in order]), dim=1).to(device)
    nbl = [neigh(c) for c in range(n_cells)]
    nbm = np.full((n_cells, 27), -1, np.int64)            # the 27-neighbourhoods as a padded matrix (ascending ids, -1 = none)
    for c, l_ in enumerate(nbl):
        nbm[c, :len(l_)] = l_
    tab = {"device": str(device), "keys": keys, "lin_of": lin_of, "neighbours": nbl, "neighbour_matrix": nbm,
           "neighbour_matrix_t": torch.from_numpy(nbm), "centers": centers,
           "diag": diag, "centers_host": centers.cpu(), "diag_host": diag.cpu()}
    try:
        scene._mcr_grid_tables = tab
    except Exception:
        pass
    return tab


class _ForeignStore:
    """Flat store (see scene._Store) of a Scene-like object that keeps one tensor per cell -- the reference's Scene."""

    def __init__(self, scene, keys, device):
        from .. import ops as _ops
        cells = [scene.cells[k] for k in keys]
        lens = [int(c.cell_pts.shape[0]) for c in cells]
        self.off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        self.off_dev = _ops.h2d(self.off, torch.int64, device)
        self.pts = torch.cat([c.cell_pts for c, n in zip(cells, lens) if n] + [torch.zeros(0, 3, device=device)]).float().contiguous()
        fd = getattr(scene, "feature_dim", 0)
        self.fts = (torch.cat([c.cell_features.reshape(n, -1) for c, n in zip(cells, lens) if n] + [torch.zeros(0, max(fd, 1), device=device)])
                    .float().contiguous() if fd > 0 else None)


def _store_of(scene, device):
    if hasattr(scene, "flat_store"):
        return scene.flat_store()
    return _ForeignStore(scene, _grid_tables(scene, device)["keys"], device)


def _grid_consts(scene, device):
    """x_min | x_max | step as 9 fp32 on the device (mcr_cell_keys' grid_consts) and the grid shape."""
    if hasattr(scene, "_consts"):
        return scene._consts(device)["gc"], (scene.grid_l, scene.grid_w, scene.grid_h)
    tab = _grid_tables(scene, device)
    if "gc" not in tab:
        step = torch.stack([torch.as_tensor(v, dtype=torch.float32).reshape(()) for v in (scene.l, scene.w, scene.h)])
        tab["gc"] = torch.cat((scene.x_min.reshape(3).float().cpu(), scene.x_max.reshape(3).float().cpu(), step.cpu())).to(device).contiguous()
    return tab["gc"], (scene.grid_l, scene.grid_w, scene.grid_h)


def _field_select(proxy_scene, device, use_supervision_occ_mask=True, pending=None):
    """Device part of the field pass's selection (ops.field_select); `pending`: a fill of the proxy scene whose host part has not
    run yet (macarons_nbv_decision reads both count tables back together)."""
    ps = proxy_scene
    st = _store_of(ps, device)
    if st.fts is None:
        raise ValueError("the proxy scene's cells must carry the proxy indices as feature (feature_dim >= 1)")
    gc, grid = _grid_consts(ps, device)
    if grid[0] * grid[1] * grid[2] > 1023:              # the counting sort of csrc/scene.hip: one LDS counter per cell (Scene.MAX_CELLS)
        raise NotImplementedError(f"the fused field pass handles grids of up to 1023 cells, this one has {grid[0] * grid[1] * grid[2]} "
                                  f"({grid[0]} x {grid[1]} x {grid[2]}); the reference's scenes use 18 .. 72")
    return ops.field_select(ps.proxy_points, ps.proxy_supervision_occ, ps.out_of_field, ps.proxy_proba, st.fts, int(st.off[-1]), st.off_dev,
                            gc, grid, use_supervision_occ_mask, pending)


def _job_groups(cloud_sizes):
    """Consecutive groups of (cell, chunk) jobs with about the same amount of hidden draws (a job draws ~2.3 indices per surface point
    of its cloud).  The batched draws (torch.ops.macarons.scone_occ_draws: prefix-only Fisher-Yates, the engine advanced over the rest)
    cost ~0.4 ms per million indices, less than the ~40 launches an extra group repeats: ONE group below 3 M indices (the bench scene
    draws 1 M: 7.8 / 8.4 / 9.0 ms for 1 / 2 / 3 groups), at most three.  env MCR_FIELD_GROUPS=n forces n."""
    import os
    J = len(cloud_sizes)
    work = np.cumsum(np.asarray(cloud_sizes, np.float64))
    forced = os.environ.get("MCR_FIELD_GROUPS")
    G = int(forced) if forced else int(min(3, max(1, round(2.3 * work[-1] / 3e6))))
    G = max(1, min(G, J))
    cuts = [0] + [int(np.searchsorted(work, work[-1] * g / G)) + 1 for g in range(1, G)] + [J]
    cuts = sorted(set(min(max(c, 0), J) for c in cuts))
    return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


_vh_matrix_t_cache = {}


def _vh_matrix_t(params, device):
    """[n_bins, 64] on the device: the constant matrix of compute_view_harmonics (scone_utils.py:953-958), transposed for the fused
    row kernel; built once per (degree, lattice, device)."""
    from . import scone_utils as su
    key = (params.harmonic_degree, params.view_state_n_elev, params.view_state_n_azim, str(device))
    m = _vh_matrix_t_cache.get(key)
    if m is None:
        base, h_polar, _ = su.get_all_harmonics_under_degree(params.harmonic_degree, params.view_state_n_elev, params.view_state_n_azim, device)
        m = su._view_harmonics_matrix(base, h_polar, params.view_state_n_elev, params.view_state_n_azim).t().contiguous()
        _vh_matrix_t_cache[key] = m
    return m


_NATIVE_FIELD_JOBS = []


def _native_field_jobs():
    """Is torch.ops.macarons.field_jobs there (the C++ extension built)?  MCR_NATIVE_FIELD_JOBS=0: the numpy restatement (A/B, tests)."""
    if not _NATIVE_FIELD_JOBS:
        import os
        ok = os.environ.get("MCR_NATIVE_FIELD_JOBS", "1") != "0"
        if ok:
            try:
                from .. import torch_ops  # noqa: F401
                ok = hasattr(torch.ops.macarons, "field_jobs")
            except Exception:
                ok = False
        _NATIVE_FIELD_JOBS.append(ok)
    return _NATIVE_FIELD_JOBS[0]


def _field_prepare_begin(params, proxy_scene, prediction_camera, device):
    """Start _field_prepare's geometry on the extension's worker thread (torch.ops.macarons.field_prepare_async: a C++ thread, no
    GIL) and return the ticket _field_prepare collects, or None where the inline form applies (no host matrix, opt-out
    MCR_FIELD_PREPARE_ASYNC=0, extension without the op)."""
    if not (torch.is_tensor(prediction_camera) and prediction_camera.device.type == "cpu" and _native_field_jobs()
            and os.environ.get("MCR_FIELD_PREPARE_ASYNC", "1") != "0" and hasattr(torch.ops.macarons, "field_prepare_async")):
        return None
    from . import scone_utils as su
    tab = _grid_tables(proxy_scene, device)
    key = (params.view_state_n_elev, params.view_state_n_azim)
    if key not in su._REF_DIRECTIONS:
        su.view_space_bin_permutation(torch.eye(3), *key)
    return torch.ops.macarons.field_prepare_async(prediction_camera.detach().to(torch.float32).reshape(4, 4), tab["centers_host"],
                                                  tab["diag_host"], su._REF_DIRECTIONS[key], float(params.prediction_neighborhood_size),
                                                  *key)


def _field_prepare(params, proxy_scene, prediction_camera, device, ticket=None):
    """Host work of the field pass that does not depend on the selection's counts (so that macarons_nbv_decision can do it while the
    GPU still works towards the read-back): the world->view matrix on the host, every cell's prediction-box transform
    (:1468-1478), the bin permutation of move_view_state_to_view_space (:863-931)."""
    from . import scone_utils as su
    tab = _grid_tables(proxy_scene, device)
    Mv_any = _world_to_view_matrix(prediction_camera)
    Mv_host = Mv_any.detach().to("cpu", torch.float32).reshape(4, 4)    # (a device matrix is read back here)
    cw, dg = tab["centers_host"], tab["diag_host"]
    n = cw.shape[0]
    if ticket is not None:
        xf_all_t, perm_t = torch.ops.macarons.field_prepare_wait(ticket)
        return {"tab": tab, "xf_all": xf_all_t.numpy(), "perm": perm_t.numpy(), "xf_all_t": xf_all_t, "perm_t": perm_t,
                "vh_mt": _vh_matrix_t(params, device)}
    if torch.is_tensor(prediction_camera) and _native_field_jobs() and hasattr(torch.ops.macarons, "field_prepare"):
        # the same ATen operators from ONE C++ call (bit-identical; tests/test_draws_cpu.py)
        su.view_space_bin_permutation(torch.eye(3), params.view_state_n_elev, params.view_state_n_azim) \
            if (params.view_state_n_elev, params.view_state_n_azim) not in su._REF_DIRECTIONS else None
        xf_all_t, perm_t = torch.ops.macarons.field_prepare(Mv_host, cw, dg, su._REF_DIRECTIONS[(params.view_state_n_elev, params.view_state_n_azim)],
                                                            float(params.prediction_neighborhood_size), params.view_state_n_elev,
                                                            params.view_state_n_azim)
        return {"tab": tab, "xf_all": xf_all_t.numpy(), "perm": perm_t.numpy(), "xf_all_t": xf_all_t, "perm_t": perm_t,
                "vh_mt": _vh_matrix_t(params, device)}
    cen_h = (torch.cat((cw, torch.ones(n, 1)), 1) @ Mv_host)[:, :3]
    inv_h = (1.0 / (params.prediction_neighborhood_size * dg)).float()
    xf_all_t = torch.cat((Mv_host.reshape(1, 16).expand(n, -1), cen_h, inv_h.view(n, 1)), 1).contiguous()
    perm_t = su.view_space_bin_permutation((Mv_host[:3, :3].contiguous() if torch.is_tensor(prediction_camera) else prediction_camera),
                                           params.view_state_n_elev, params.view_state_n_azim, device).to(torch.int32)
    return {"tab": tab, "xf_all": xf_all_t.numpy(), "perm": perm_t.numpy(), "xf_all_t": xf_all_t, "perm_t": perm_t,
            "vh_mt": _vh_matrix_t(params, device)}


def compute_scene_occupancy_probability_field(params, macarons, camera, surface_scene, proxy_scene, device,
                                              use_supervision_occ_mask=True, prediction_camera=None,
                                              use_supervision_occ_instead_of_predicted=False, chunk=20000, ragged_perms=None,
                                              group=None, record=None, _selection=None, _between=None, _after=None):
    """Occupancy probability of every proxy point the cameras have seen (macarons_utils.py:1395-1540), as ONE batched pass.

    Upstream walks the grid cells that hold seen proxy points from Python: per cell it gathers the surface points of the 27-cell
    neighbourhood and the cell's registered proxy points, moves both to the prediction camera's view space (centred on the cell,
    scaled by prediction_neighborhood_size x the cell diagonal), rotates the view states into that frame, and calls the occupancy
    network in chunks of 20 000 queries.  Here the cells are SEGMENTS of flat device arrays: one call selects and groups the proxy
    points by cell (ops.field_select: a counting sort), the host learns two integers per cell in a single read-back and lists the
    (cell, chunk) jobs, one call builds every job's surface cloud, queries and view harmonics in its prediction space
    (ops.field_build), and all jobs go through SconeOcc.forward_ragged together; the hidden draws of the network are made job by
    job on the CPU generator, i.e. exactly the draws of the cell loop.
    Returns (X_world [N,3], view_harmonics [N,64], occ_probs [N,1]) in upstream's order (cells in lexicographic order, points by
    index, then the never-seen points with their stored probability) and updates proxy_scene.proxy_proba in place.
    `surface_scene` / `proxy_scene`: macarons_amd.utility.scene.Scene or objects with the reference Scene's attributes;
    `prediction_camera`: a PyTorch3D-like camera, or the [4,4] world->view matrix.
    `group` (torch.distributed, ranks holding replicas of both scenes; None = local): the T query rows of all jobs are
    block-partitioned over the ranks (every (cell, chunk) job is independent, and so is every query of a job given the
    job's cloud and draws), each rank runs the jobs its rows belong to, the occupancies (4 B per proxy point) are all-gathered; the
    hidden draws of ALL jobs are rank 0's, in job order, in one broadcast -- the result is bit for bit the 1-rank field.  `record`
    (dict): receives the draws used (`ragged_perms`) so that a caller can repeat the pass.  `_selection` / `_between` / `_after` (macarons_nbv_decision): a selection whose counts are already on the host (with the
    count-independent host work, _field_prepare); host work to run once the first launches of the occupancy pass are queued and BEFORE
    the network's draws (the fill's draws: upstream's order on the CPU generator); host work to run once the whole pass is queued
    (the fill's gather: the GPU is busy meanwhile)."""
    from .. import dist as mdist
    world, rank = mdist.group_world_rank(group)            # group=None: local, whatever process groups exist
    ps, ss = proxy_scene, surface_scene
    gl, gw, gh = ps.grid_l, ps.grid_w, ps.grid_h
    n_cells = gl * gw * gh
    nh = params.n_harmonics
    if prediction_camera is None:
        if camera is None:
            raise NameError("Both camera and prediction_camera are equal to None.")
        prediction_camera = camera.fov_camera_0
    # ---- selection: which proxy points take part, grouped by the cell that stores them (:1428-1442); ONE read-back of the counts
    if _selection is None:
        sel = _field_select(ps, device, use_supervision_occ_mask)
        prep = _field_prepare(params, ps, prediction_camera, device)
        hostc = sel.counts.cpu().numpy()
    else:
        sel, hostc, prep = _selection
    visit, counts = hostc[:n_cells], hostc[n_cells + 1:2 * n_cells + 1]
    sel_off = hostc[2 * n_cells + 2:3 * n_cells + 4]
    n_oof = int(hostc[3 * n_cells + 4])
    tab = prep["tab"]                                    # static per scene: cells in linear-id order, their centres / diagonals, 27-neighbourhoods
    keys, lin_of = tab["keys"], tab["lin_of"]
    # ---- host: which cells run, their surface neighbourhoods (sizes are host numbers: no read-back), the (cell, chunk) jobs
    s_st = _store_of(ss, device)
    s_keys = keys if set(ss.cells.keys()) == set(keys) else sorted(ss.cells.keys(), key=lambda k: lin_of.get(k, 0))
    if len(s_keys) == n_cells:
        s_len, s_start = np.diff(s_st.off), s_st.off[:-1]
    else:                                               # a surface scene on another grid: by key
        s_len = np.zeros(n_cells, np.int64); s_start = np.zeros(n_cells, np.int64)
        for i_, k in enumerate(s_keys):
            s_len[lin_of[k]] = s_st.off[i_ + 1] - s_st.off[i_]; s_start[lin_of[k]] = s_st.off[i_]
    nbm = tab["neighbour_matrix"]                        # [n_cells, 27], -1 padded
    native = len(s_keys) == n_cells and _native_field_jobs()
    if native:
        # the job tables by ONE C++ call (torch.ops.macarons.field_jobs: the loop form of the numpy code below, same tables element for
        # element) -- this host work sits between the decision's read-back and the first launch behind it, with the GPU idle
        raw_t, meta_t = torch.ops.macarons.field_jobs(torch.from_numpy(np.ascontiguousarray(hostc)), torch.from_numpy(np.ascontiguousarray(s_st.off)),
                                                      tab["neighbour_matrix_t"], prep["xf_all_t"], prep["perm_t"], n_cells, int(chunk),
                                                      int(params.k_for_knn))
        meta = meta_t.numpy()
        J, n_seg, T, tot = int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])
        job_q, job_m = meta[5:5 + J], meta[5 + J:5 + 2 * J]
        q_start, m_start = meta[5 + 2 * J:6 + 3 * J], meta[6 + 3 * J:7 + 4 * J]
    else:
        nb_len = np.where(nbm >= 0, s_len[np.maximum(nbm, 0)], 0)          # surface points of every neighbour cell
        m_cell = nb_len.sum(1)
        run = (np.asarray(visit) != 0) & (m_cell > 2 * 2 * params.k_for_knn) & (np.asarray(counts) > 0)       # :1455-1456
        cells_run = np.nonzero(run)[0]
        n_chunks = -(-np.asarray(counts)[cells_run] // chunk)
        job_cell = np.repeat(cells_run, n_chunks)                          # one job per (cell, chunk of <= 20 000 queries)
        J = int(job_cell.size)
        T = tot = 0
    P = ps.proxy_points.shape[0]
    if J and not native:
        lo = (np.arange(J) - np.repeat(np.cumsum(n_chunks) - n_chunks, n_chunks)) * chunk
        job_q = np.minimum(chunk, np.asarray(counts)[job_cell] - lo).astype(np.int64)
        job_m = m_cell[job_cell].astype(np.int64)
        q_start = np.concatenate(([0], np.cumsum(job_q)))
        m_start = np.concatenate(([0], np.cumsum(job_m)))
        T, tot = int(q_start[-1]), int(m_start[-1])
        jt = np.stack((sel_off[job_cell] + lo, q_start[:-1], m_start[:-1], np.zeros(J, np.int64)), 1).astype(np.int64)
        # surface segments of every job: its cell's non-empty neighbours, in ascending cell order
        seg_len = nb_len[job_cell]                                         # [J, 27]
        keep = seg_len > 0
        seg_job = np.nonzero(keep)[0]
        seg_l = seg_len[keep]
        seg_dst = np.cumsum(seg_l) - seg_l
        st_ = np.stack((s_start[nbm[job_cell][keep]], seg_dst, seg_job, np.zeros(seg_l.size, np.int64)), 1).astype(np.int64)
    N_out = T + n_oof
    X_world = torch.empty((N_out, 3), dtype=torch.float32, device=device)
    view_harmonics = torch.empty((N_out, nh), dtype=torch.float32, device=device)
    occ_probs = torch.empty((N_out, 1), dtype=torch.float32, device=device)
    rows = None
    if J:
        # ---- ONE upload: job table, segment table, per-job transform (world->view matrix, box centre in view space, 1 / (neighbourhood
        # size x cell diagonal), :1468-1478), the bin permutation of move_view_state_to_view_space (:863-931)
        if native:
            tables = ops.h2d(raw_t, torch.uint8, device)
        else:
            xf = np.ascontiguousarray(prep["xf_all"][job_cell])
            raw = np.concatenate((jt.reshape(-1).view(np.uint8), st_.reshape(-1).view(np.uint8), xf.reshape(-1).view(np.uint8),
                                  prep["perm"].view(np.uint8)))
            tables = ops.h2d(raw, torch.uint8, device)
            n_seg = int(st_.shape[0])
        perm_d = tables[32 * (J + n_seg) + 80 * J:].view(torch.int32)
        rows, row_job, X_q, pc_all = ops.field_build(tables, J, n_seg, sel, ps.proxy_points, s_st.pts, ps.view_states, perm_d,
                                                     prep["vh_mt"], T, tot, X_world, view_harmonics)
        vh = view_harmonics[:T]
        occ_out = occ_probs[:T]
        sizes_m, sizes_q = job_m.tolist(), job_q.tolist()
        # ---- occupancy of all jobs
        if use_supervision_occ_instead_of_predicted:
            if _between is not None:
                _between()
            occ_out.copy_(ps.proxy_supervision_occ[rows.long()])
        else:
            occ_net = getattr(macarons, "occupancy", macarons)
            if hasattr(occ_net, "forward_ragged") and mdist.exchange_on(group):
                if _between is not None:
                    _between()
                if ragged_perms is None:                # rank 0 draws for every job, in job order (what the 1-rank pass draws)
                    ragged_perms = _broadcast_job_perms(occ_net, sizes_m, device, group, rank)
                t0, t1 = mdist.shard_range(T, rank, world)
                mine = [j for j in range(J) if q_start[j] < t1 and q_start[j + 1] > t0]
                if mine:
                    q_l = [int(min(q_start[j + 1], t1) - max(q_start[j], t0)) for j in mine]
                    p0, p1 = int(m_start[mine[0]]), int(m_start[mine[-1] + 1])
                    draws_l = (_slice_index_arrays(occ_net, ragged_perms, sizes_m, mine[0], mine[-1] + 1) if isinstance(ragged_perms, dict)
                               else [ragged_perms[j] for j in mine])
                    kw = {"index_arrays": draws_l} if isinstance(draws_l, dict) else {"perms": draws_l}
                    occ_l = occ_net.forward_ragged(pc_all[p0:p1].contiguous(), [sizes_m[j] for j in mine], X_q[t0:t1].contiguous(),
                                                   vh[t0:t1].contiguous(), q_l, **kw).view(-1, 1)
                else:                                   # empty row shard (T < world): no kernels, the all-gather is joined
                    occ_l = torch.zeros(0, 1, dtype=torch.float32, device=device)
                occ_out.copy_(mdist.allgather_rows(occ_l, T, group))
            elif hasattr(occ_net, "forward_ragged"):
                # The jobs run in G consecutive GROUPS when their hidden draws are made here on the CPU generator (~1 ms per 400k
                # permuted indices, inherent to the reference's stream): the first launches of EVERY group (which need no draw) are queued
                # at once, then, while the GPU works on group g, the host draws for group g + 1 -- only the first group's draws pass with
                # the GPU short of work.  Every job computes what it computes alone (launch shapes are chosen per cloud), so the grouping
                # does not change a bit of the result; the draws stay in job order.
                given = ragged_perms["groups"] if (isinstance(ragged_perms, dict) and "groups" in ragged_perms) else None
                if given is not None:
                    groups = [(j0, j1) for j0, j1, _ in given]
                elif ragged_perms is None:
                    groups = _job_groups(sizes_m)
                else:
                    groups = [(0, J)]
                hdls = []
                for gi, (j0, j1) in enumerate(groups):
                    t0, t1, p0, p1 = int(q_start[j0]), int(q_start[j1]), int(m_start[j0]), int(m_start[j1])
                    whole = (j0, j1) == (0, J)
                    hdls.append(occ_net.forward_ragged_begin(pc_all if whole else pc_all[p0:p1], sizes_m[j0:j1], X_q if whole else X_q[t0:t1],
                                                             vh if whole else vh[t0:t1], sizes_q[j0:j1], row_job=row_job if whole else None,
                                                             arena="scone_occ_ragged" if gi == 0 else f"scone_occ_ragged_g{gi}"))
                if _between is not None:
                    _between()                          # phase 1 is queued; the fill's CPU draws come first, as upstream (Cell.fill, then the pass)
                used = []
                for gi, (j0, j1) in enumerate(groups):
                    t0, t1 = int(q_start[j0]), int(q_start[j1])
                    src = given[gi][2] if given is not None else ragged_perms
                    kw = {"index_arrays": src} if isinstance(src, dict) else {"perms": src}
                    occ_net.forward_ragged_finish(hdls[gi], out=occ_out if (j0, j1) == (0, J) else occ_out[t0:t1], **kw)
                    used.append((j0, j1, occ_net.last_ragged_perms))
                ragged_perms = used[0][2] if len(used) == 1 else {"groups": used}
            else:                                       # any other module with the reference's call signature: job by job
                if _between is not None:
                    _between()
                r0, p0 = 0, 0
                for q, m in zip(sizes_q, sizes_m):
                    occ_out[r0:r0 + q] = macarons(mode='occupancy', partial_point_cloud=pc_all[p0:p0 + m][None], proxy_points=X_q[r0:r0 + q][None],
                                                  view_harmonics=vh[r0:r0 + q][None]).view(-1, 1)
                    r0, p0 = r0 + q, p0 + m
            if record is not None:
                record["ragged_perms"] = ragged_perms
    elif _between is not None:
        _between()
    # ---- :1525 (the stored probabilities of the visited points), then the field's tail: the never-seen points in ascending order with
    # their stored probability and zero harmonics (:1531-1537)
    if n_oof:
        view_harmonics[T:].zero_()
    ops.field_finish(rows, occ_probs, T, ps.proxy_proba, sel, n_oof, ps.proxy_points, X_world[T:], occ_probs[T:])
    if _after is not None:
        _after()
    return X_world, view_harmonics, occ_probs


def _broadcast_job_perms(occ_net, cloud_sizes, device, group, rank):
    """The hidden draws of SconeOcc for J jobs (three index tensors per job, SconeOcc.draw_perms), made by rank 0 in job order on
    its CPU generator -- exactly what a 1-rank pass draws -- and handed to every rank in ONE broadcast.  Every rank knows the
    sizes (they follow from the cloud sizes), so the other ranks only allocate."""
    from .. import dist as mdist
    lens = []
    for m_ in cloud_sizes:
        sz = occ_net.scale_sizes(int(m_))
        lens.append([min(int(m_), occ_net.seq_len)] + sz[1:])
    total = sum(sum(l_) for l_ in lens)
    if rank == 0:
        drawn = [occ_net.draw_perms(int(m_)) for m_ in cloud_sizes]
        flat = torch.cat([p_.reshape(-1) for job in drawn for p_ in job]).to(torch.int64)
        buf = ops.h2d(flat, torch.int64, device)
    else:
        buf = torch.empty(total, dtype=torch.int64, device=device)
    mdist.broadcast(buf, 0, group)
    host = buf.cpu()
    out, o = [], 0
    for l_ in lens:
        job = []
        for n_ in l_:
            job.append(host[o:o + n_]); o += n_
        out.append(job)
    return out


def _slice_index_arrays(occ_net, ia, cloud_sizes, j0, j1):
    """The index arrays of jobs j0 .. j1-1 out of those of all jobs, re-based to the sub-list's own row offsets."""
    Lg = occ_net.seq_len
    sz = [occ_net.scale_sizes(int(m_)) for m_ in cloud_sizes]
    c0 = int(sum(s_[0] for s_ in sz[:j0]))
    a1, b1 = int(sum(s_[1] for s_ in sz[:j0])), int(sum(s_[1] for s_ in sz[:j1]))
    a2, b2 = int(sum(s_[2] for s_ in sz[:j0])), int(sum(s_[2] for s_ in sz[:j1]))
    return {"g_idx": ia["g_idx"][j0 * Lg:j1 * Lg] - c0, "g_len": ia["g_len"][j0:j1], "idx1": ia["idx1"][a1:b1] - c0,
            "idx2": ia["idx2"][a2:b2] - a1, "off1": ia["off1"][j0:j1 + 1] - a1, "off2": ia["off2"][j0:j1 + 1] - a2}


def compute_occupancy_probability(macarons, pc, X, view_harmonics, mask=None, max_points_per_pass=20000):
    """macarons_utils.py:1194-1231: chunked occupancy inference through Macarons.forward(mode='occupancy')."""
    n_clouds, n_sample = pc.shape[0], X.shape[1]
    p = max_points_per_pass // n_clouds
    preds = [torch.zeros(n_clouds, 0, 1, device=X.device)]
    for low in range(0, n_sample, p):
        up = min(low + p, n_sample)
        preds.append(macarons(mode='occupancy', partial_point_cloud=pc, proxy_points=X[:, low:up].contiguous(),
                              view_harmonics=view_harmonics[:, low:up].contiguous()).view(n_clouds, up - low, -1))
    return torch.cat(preds, dim=1)
