"""Mirror of the MACARONS-regime scoring helpers of macarons/utility/macarons_utils.py on the MI355X kernels:
  get_distance_factor :1741-1765, get_distance_factor_threshold :1768-1776, get_distance_factor_smooth :1779-1788
  predict_coverage_gain_for_single_camera :1580-1738  (here batched over the <= 30 neighbour cameras)
PyTorch3D camera objects stay outside: cameras are passed as the 40-float records of mcr_points_in_fov
(M_view, M_proj, ndc bounds, centre, range) and a prediction-view matrix.
"""
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
                                      distance_th=17., samples=None, smooth=False, return_parts=False, record=None):
    """The per-neighbour-camera scoring loop of testers/scene.py:434-454 around
    predict_coverage_gain_for_single_camera (macarons_utils.py:1580-1738), for K cameras AT ONCE and without a host
    synchronisation (the reference runs one SconeVis forward and reads a count back per camera):
      frustum mask (all K at once) -> occupancy-weighted sampling inside each frustum -> prediction-view space ->
      SconeVis -> per-point visibility gains (C = 1) x distance factor -> mean x sum(occ in frustum).
    X_world [P,3], proxy_view_harmonics [P,64], occ_probs [P,1], cameras [K,40], X_cam_world [K,3],
    prediction_view_matrices [K,4,4] (world -> prediction-camera view, row-vector).  distance_th / smooth select the distance
    factor (params.distance_factor_th: a number -> threshold factor; None -> sensor_distance_threshold(...), smooth=False;
    'smooth' -> sensor_distance_threshold(...), smooth=True).  Returns gains [K] (and, with return_parts, the per-camera lists of
    factored per-point gains [N] and sampled world points [N,4])."""
    K = cameras.shape[0]
    dev = X_world.device
    S = seq_len
    mask = ops.points_in_fov(X_world, cameras)                                            # :1603
    occ_k = ops.fov_mask_occ(mask, occ_probs.reshape(-1).contiguous())                    # :1606-1613 folded into the sampler
    # ---- sampling inside every frustum (:1624): K distributions over the ONE shared point set in one launch sequence, nothing
    # read back (padded rows, counts on the device).  Uniforms: ONE draw of [K, S] from the device generator (upstream draws
    # torch.rand(S, 1) inside every per-camera call; 30 launches of a 2 us kernel are 0.35 ms of a host-bound decision -- callers that
    # need particular uniforms pass `samples`).
    if samples is None:
        u = torch.rand(K, S, device=dev)
    elif torch.is_tensor(samples):
        u = samples.to(dev).reshape(K, S).float()
    else:
        u = torch.stack([torch.as_tensor(x, device=dev).reshape(-1) for x in samples]).float()
    if record is not None:
        record["samples"] = u                                                               # (a re-run must see the same uniforms)
    res, res_h, inv, _, nu, vol = ops.sample_proxy_batched(X_world, occ_k, proxy_view_harmonics, u.contiguous(), min_occ)
    vol = vol.float()                                                                      # [K,S,4] [K,S,64] [K,S]; [K] int32, [K]
    # ---- prediction box: centre of the sampled points' bounding box, in the prediction camera's view space (:1631-1641)
    valid = (torch.arange(S, device=dev)[None, :] < nu[:, None])[..., None]               # [K,S,1]
    xyz = res[..., :3]
    hi = torch.where(valid, xyz, torch.full_like(xyz, float("-inf"))).amax(dim=1)
    lo = torch.where(valid, xyz, torch.full_like(xyz, float("inf"))).amin(dim=1)
    center_w = torch.where((nu > 0)[:, None], (hi + lo) / 2., torch.zeros_like(hi))       # empty frustum: any finite centre
    Mv = prediction_view_matrices.to(device=dev, dtype=torch.float32).contiguous()
    center = torch.bmm(torch.cat((center_w, torch.ones(K, 1, device=dev)), 1)[:, None, :], Mv)[:, 0, :3].contiguous()
    pts = res.clone()
    cam4 = torch.cat((X_cam_world.reshape(K, 3), torch.ones(K, 1, device=dev)), 1).contiguous()
    inv_diag = torch.full((K,), 1.0 / prediction_box_diag, dtype=torch.float32, device=dev)
    ops.transform_points_batched_(pts, Mv, center, inv_diag)                               # :1647-1650, all cameras in one launch
    ops.transform_points_batched_(cam4.view(K, 1, 4), Mv, center, inv_diag)                # :1655-1659
    # ---- ONE SconeVis forward over the K padded clouds (:1664), ONE scorer launch (C = 1 per cloud, :1683), ONE gain launch
    harm = visibility_model(pts, view_harmonics=res_h, lengths=nu)
    gi = inv[..., None]
    pts_mc = torch.gather(pts, 1, gi.expand(-1, -1, 4)).contiguous()                      # MC duplicates (:1668-1671)
    harm_mc = torch.gather(harm, 1, gi.expand(-1, -1, 64)).contiguous()
    world = torch.gather(res, 1, gi.expand(-1, -1, 4)).contiguous()
    vis = ops.sh_visibilities(pts_mc, harm_mc, cam4[:, None, :3].contiguous(), True).view(K, S)
    gains = ops.macarons_gain_(vis, world, X_cam_world.reshape(K, 3).contiguous(), vol, distance_th, smooth)     # :1699-1704
    gains = torch.where(nu > 0, gains, torch.zeros_like(gains))                           # empty frustum: gain 0 (:1707-1736)
    if return_parts:
        n_host = nu.tolist()
        return (gains, [vis[k] if n_host[k] > 0 else None for k in range(K)],
                [world[k] if n_host[k] > 0 else None for k in range(K)])
    return gains


class SceneCamera:
    """What one MACARONS decision reads of the reference's Camera / PyTorch3D objects (which stay outside the kernels):
    the 40-float record of mcr_points_in_fov (world->view matrix, full projection matrix, NDC bounds, centre, sensor range), the
    camera centre X_cam [1,3], zfar and the field of view in degrees."""

    def __init__(self, record, X_cam, zfar, fov=60.0):
        self.record, self.X_cam, self.zfar = record.reshape(40).contiguous(), X_cam.reshape(1, 3).contiguous(), float(zfar)
        self.fov = torch.tensor([float(fov)])

    @property
    def M_view(self):
        return self.record[:16].view(4, 4)


def macarons_nbv_decision(params, macarons, proxy_scene, surface_scene, camera, depth, depth_mask, neighbor_records, X_neighbors,
                          device, samples=None, return_signed_distances=False, range_guard=True, group=None, perm_source="host"):
    """One next-best-view decision of the MACARONS loop after the depth map of the current pose is known -- the body of
    testers/scene.py:391-454 (everything between the depth network and the move to the chosen pose):
      1. proxy points in the current frustum (Camera.get_points_in_fov :391), registered in the proxy grid (:394-395);
      2. signed distances to the depth map, view-state OR, supervision occupancy, out-of-field flags (:397-415) -- one fused pass;
      3. surface features reset (:418); occupancy-probability field over the seen cells (:421-425);
      4. coverage gain of every valid neighbour pose (:434-450), all cameras in one launch sequence; first strict maximum (:452-454).
    camera: SceneCamera of the current pose (it is also the prediction camera, fov_camera_0 of :305); depth [H,W] (+ optional
    leading/trailing singleton dims), depth_mask like depth; neighbor_records [K,40], X_neighbors [K,3].
    `group` (torch.distributed; every rank holds replicas of both scenes and calls with the same arguments): SURVEY §8e -- the query
    rows of the occupancy field and the K neighbour cameras are block-partitioned over the ranks, the occupancies (4 B per proxy
    point) and one 8-byte (gain, index) record per rank are all-gathered, the hidden draws (Cell.fill subsets, SconeOcc's
    down-samples, the sampling uniforms) are rank 0's; the cheap state updates run replicated.  Bit for bit the 1-rank decision;
    `gains` then holds this rank's cameras only (`cam_range`).
    perm_source: "host" (default) draws the hidden permutations (Cell.fill's subsets, SconeOcc's down-samples) with torch.randperm on
    the CPU generator in upstream's order -- what the reference goldens pin; "device" (opt-in, production) draws them on the GPU in
    two segmented sorts (statistically the same, a different stream): ~190 host draws = 2.7 ms of CPU time per decision less.
    Returns dict(next_idx (device int64: index into the neighbour list), gains [K], fov_mask [P] bool, X_world, view_harmonics,
    occ_probs).  The scene objects are updated in place like upstream.  Host synchronisations: the cell counts of the occupancy-field
    pass (cell bookkeeping on the host, as upstream), fill_cells' one, and -- range_guard=True -- the range flag of the fp16-split
    path, read once at the end (range_guard=False: the caller checks macarons.occupancy.range_flag() itself)."""
    from .. import dist as mdist
    world, rank = mdist.group_world_rank(group)            # group=None: local, whatever process groups exist
    H, W = params.image_height, params.image_width
    depth2 = depth.reshape(H, W).contiguous().float()
    dmask2 = depth_mask.reshape(H, W) if depth_mask is not None else None
    rec = ops.h2d(camera.record, torch.float32, device)
    # 1 ---- proxy points in the current field of view, registered in their grid cells with their index as feature
    fov_mask = ops.points_in_fov(proxy_scene.proxy_points, rec.view(1, 40))[0]
    # every proxy point is offered with its index as feature, the ones outside the frustum flagged invalid: compacting them first
    # (`points[mask]`) is a read-back of the count
    P_all = proxy_scene.proxy_points.shape[0]
    proxy_scene.fill_cells(proxy_scene.proxy_points, features=torch.arange(P_all, device=device, dtype=torch.float32).view(-1, 1),
                           valid=fov_mask, **({"group": group} if world > 1 else {}),
                           **({"perm_source": perm_source} if perm_source != "host" else {}))
    # 2 ---- carve with the depth map: signed distance, view states, supervision occupancy, out-of-field, one launch
    sgn = proxy_scene.update_from_depth(fov_mask, rec, ops.h2d(camera.X_cam, torch.float32, device), depth2, dmask2, fill=1.1 * camera.zfar,
                                        tol=params.carving_tolerance, return_signed_distances=return_signed_distances)
    surface_scene.set_all_features_to_value(value=1.)
    # 3 ---- occupancy probability field, in the current camera's view space; 4 ---- neighbours.
    # The range check of the fp16-split path (SconeOcc.range_guard) is DEFERRED to one read-back at the very end: checked inside
    # the occupancy pass it stalls the host until the pass has run, and the ~200 small launches of the glue behind it then start
    # on an idle GPU (kernel trace of one decision: 9 of 22 ms idle).  On a set flag steps 3-4 are repeated on the full-range
    # variant with the SAME hidden draws (cell permutations, sampling uniforms).
    Mv_field = camera.M_view                          # where the caller keeps it: a host matrix is used on the host, uploaded without a stall
    Mv = ops.h2d(Mv_field, torch.float32, device)
    K = neighbor_records.shape[0]
    th = params.distance_factor_th
    smooth = th == 'smooth'
    if th is None or smooth:
        th = sensor_distance_threshold(params, camera, surface_scene.cell_resolution)
    vis_model = macarons.visibility                 # `macarons` = the SCONE part (Macarons.scone upstream): .occupancy / .visibility
    diag = getattr(proxy_scene, "_mcr_box_diag", None)      # a constant of the scene: read back once, not once per decision
    if diag is None:
        diag = torch.linalg.norm(proxy_scene.x_max - proxy_scene.x_min).item()
        try:
            proxy_scene._mcr_box_diag = diag
        except Exception:
            pass
    occ_net = getattr(macarons, "occupancy", macarons)
    nrec, xn = ops.h2d(neighbor_records, torch.float32, device), ops.h2d(X_neighbors, torch.float32, device)

    k0, k1 = mdist.shard_range(K, rank, world)
    S = params.seq_len

    def field_and_gains(ragged_perms, smp, record):
        X_world, view_harmonics, occ_probs = compute_scene_occupancy_probability_field(params, macarons, None, surface_scene, proxy_scene,
                                                                                       device, prediction_camera=Mv_field, ragged_perms=ragged_perms,
                                                                                       group=group if world > 1 else None, record=record,
                                                                                       perm_source=perm_source)
        if world > 1:                                   # the uniforms of ALL cameras are rank 0's
            if smp is None:
                smp = torch.rand(K, S, device=device)
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
                                                  smooth=smooth, record=record)
        if record is not None and "samples" in record:
            record["samples_all"] = record["samples"]
        return X_world, view_harmonics, occ_probs, gains

    deferred = range_guard and getattr(occ_net, "range_guard", None) == "sync" and hasattr(occ_net, "forward_ragged")
    record = {}
    if deferred:
        occ_net.range_guard = "defer"
        occ_net.clear_range_flag(device)                # (exists on every rank before the pass: the all-reduce below is rank-invariant)
    try:
        X_world, view_harmonics, occ_probs, gains = field_and_gains(None, samples, record)
        fallback = None
        if deferred:
            flag = occ_net.range_flag()
            if world > 1:
                flag = mdist.all_reduce_max(flag, group)                # every rank repeats, or none
            if flag is not None and int(flag):      # the one read-back; out of the fp16 range: repeat on the full-range variant
                from .. import _lib
                import ctypes
                L = _lib.lib()
                v0 = L.mcr_get_local_pct_variant()
                L.mcr_set_local_pct_variant(ctypes.c_int(5))
                try:
                    X_world, view_harmonics, occ_probs, gains = field_and_gains(record.get("ragged_perms"), record.get("samples_all"), None)
                finally:
                    L.mcr_set_local_pct_variant(ctypes.c_int(v0))
                occ_net.clear_range_flag()
                fallback = 5
    finally:
        if deferred:
            occ_net.range_guard = "sync"
    # `if coverage_gain > max_coverage_gain` from -1: the first strict maximum (a NaN gain never wins upstream; here it would)
    if world > 1:                                       # ties -> the lowest index over all ranks = the first strict maximum
        max_gain, next_idx = mdist.allgather_best(gains.view(1, -1), k0, group)
        out = {"next_idx": next_idx[0], "max_gain": max_gain[0], "cam_range": (k0, k1)}
    else:
        rec_best = ops.best_record(gains.view(1, K), 0)
        out = {"next_idx": rec_best[0, 1].to(torch.int64), "max_gain": rec_best[0, 0]}
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
    tab = {"device": str(device), "keys": keys, "lin_of": lin_of, "neighbours": [neigh(c) for c in range(n_cells)], "centers": centers,
           "diag": diag, "centers_host": centers.cpu(), "diag_host": diag.cpu()}
    try:
        scene._mcr_grid_tables = tab
    except Exception:
        pass
    return tab


def compute_scene_occupancy_probability_field(params, macarons, camera, surface_scene, proxy_scene, device,
                                              use_supervision_occ_mask=True, prediction_camera=None,
                                              use_supervision_occ_instead_of_predicted=False, chunk=20000, ragged_perms=None,
                                              group=None, record=None, perm_source="host"):
    """Occupancy probability of every proxy point the cameras have seen (macarons_utils.py:1395-1540), as ONE batched pass.

    Upstream walks the grid cells that hold seen proxy points from Python: per cell it gathers the surface points of the 27-cell
    neighbourhood and the cell's registered proxy points, moves both to the prediction camera's view space (centred on the cell,
    scaled by prediction_neighborhood_size x the cell diagonal), rotates the view states into that frame, and calls the occupancy
    network in chunks of 20 000 queries.  Here the cells are SEGMENTS of flat device arrays: one stable sort groups the selected
    proxy points by cell, one gather builds every cell's surface cloud, one launch each transforms clouds / queries with a per-row
    cell id, one gather + one product give the view harmonics, and all (cell, chunk) jobs go through SconeOcc.forward_ragged
    together.  The host learns two integers per cell (visited?, how many selected points) in a single read-back; the hidden draws
    of the network are made job by job on the CPU generator, i.e. exactly the draws of the cell loop.
    Returns (X_world [N,3], view_harmonics [N,64], occ_probs [N,1]) in upstream's order (cells in lexicographic order, points by
    index, then the never-seen points with their stored probability) and updates proxy_scene.proxy_proba in place.
    `surface_scene` / `proxy_scene`: macarons_amd.utility.scene.Scene or objects with the reference Scene's attributes;
    `prediction_camera`: a PyTorch3D-like camera, or the [4,4] world->view matrix.
    `group` (torch.distributed, ranks holding replicas of both scenes): the T query rows of all jobs are block-partitioned over the
    ranks (every (cell, chunk) job is independent, and so is every query of a job given the job's cloud and draws), each
    rank runs the jobs its rows belong to, the occupancies (4 B per proxy point) are all-gathered; the hidden draws of ALL jobs are
    rank 0's, in job order, in one broadcast -- the result is bit for bit the 1-rank field.  `record` (dict): receives the draws
    used (`ragged_perms`) so that a caller can repeat the pass.  perm_source="device" (opt-in): SconeOcc's hidden down-samples are
    drawn on the device (SconeOcc.ragged_index_arrays_device) instead of ~3 torch.randperm calls per job on the host."""
    from . import scone_utils as su
    from .. import dist as mdist
    world, rank = mdist.group_world_rank(group)            # group=None: local, whatever process groups exist
    ps, ss = proxy_scene, surface_scene
    gl, gw, gh = ps.grid_l, ps.grid_w, ps.grid_h
    n_cells = gl * gw * gh
    P = ps.proxy_points.shape[0]
    nh = params.n_harmonics
    occ_mask = (ps.proxy_supervision_occ > 0.)[..., 0]
    seen = occ_mask & (ps.out_of_field < 1.)[..., 0]
    visit_pts = seen if use_supervision_occ_mask else (ps.out_of_field < 1.)[..., 0]
    ps.proxy_proba.masked_fill_(seen.view(-1, 1), 0.)                                            # :1431
    if prediction_camera is None:
        if camera is None:
            raise NameError("Both camera and prediction_camera are equal to None.")
        prediction_camera = camera.fov_camera_0
    Mv_any = _world_to_view_matrix(prediction_camera)
    Mv_host = Mv_any.detach().to(torch.float32) if Mv_any.device.type == "cpu" else None     # (a host matrix stays usable on the host)
    Mv = ops.h2d(Mv_host, torch.float32, device).contiguous() if Mv_host is not None else Mv_any.to(device=device, dtype=torch.float32).contiguous()
    # ---- per proxy point: the cell its coordinates fall in (which cells are visited, :1434) and the cell whose store holds it
    cell_by_pos = (ps.linear_cell_ids(ps.proxy_points) if hasattr(ps, "linear_cell_ids")
                   else _lin(ps.get_cells_for_each_pt(ps.proxy_points), gw, gh))
    tab = _grid_tables(ps, device)                       # static per scene: cells in linear-id order, their centres / diagonals, 27-neighbourhoods
    keys, lin_of = tab["keys"], tab["lin_of"]
    stored_cell = torch.full((P,), -1, dtype=torch.int64, device=device)
    filled = [k for k in keys if ps.cells[k].cell_pts.shape[0] > 0]
    if filled:                                           # every stored index -> its cell, three launches for the whole grid
        idx_all = torch.cat([ps.cells[k].cell_features[:, 0] for k in filled]).long()
        lens = torch.tensor([ps.cells[k].cell_pts.shape[0] for k in filled], dtype=torch.int64)
        lins = torch.tensor([lin_of[k] for k in filled], dtype=torch.int64)
        stored_cell[idx_all] = ops.h2d(torch.repeat_interleave(lins, lens), torch.int64, device)
    sel = stored_cell >= 0
    if use_supervision_occ_mask:
        sel = sel & occ_mask
    big = torch.full_like(stored_cell, n_cells)
    visit = torch.zeros(n_cells + 1, dtype=torch.int64, device=device).scatter_(0, torch.where(visit_pts, cell_by_pos, big), 1)
    counts = torch.zeros(n_cells + 1, dtype=torch.int64, device=device).scatter_add_(0, torch.where(sel, stored_cell, big),
                                                                                   torch.ones_like(stored_cell))
    # the one read-back of the pass; the number of never-seen points rides along (the tail of the field is their compaction: with the
    # count known here, it is built without a second read-back -- torch.nonzero after the occupancy pass stalled the host until the
    # pass had run, and the launches of everything behind it then started on an idle GPU)
    oof_mask = (ps.out_of_field > 0.)[..., 0]
    n_oof_t = torch.zeros(n_cells + 1, dtype=torch.int64, device=device)
    n_oof_t[0] = oof_mask.sum()
    host3 = torch.stack((visit, counts, n_oof_t)).cpu().numpy()
    host, n_oof = host3[:2, :n_cells], int(host3[2, 0])
    # ---- host: which cells run, their surface neighbourhoods (sizes are tensor shapes: no read-back), the (cell, chunk) jobs
    s_keys = keys if set(ss.cells.keys()) == set(keys) else sorted(ss.cells.keys(), key=lambda k: lin_of.get(k, 0))
    s_len = {lin_of[k]: int(ss.cells[k].cell_pts.shape[0]) for k in s_keys}
    s_start, o = {}, 0
    for k in s_keys:
        s_start[lin_of[k]] = o
        o += s_len[lin_of[k]]

    neighbours = tab["neighbours"].__getitem__
    jobs, seg_src, seg_len = [], [], []                 # job = (cell, number of queries); surface segments in job order
    valid_cell = torch.zeros(n_cells + 1, dtype=torch.bool)
    for c in range(n_cells):
        if not host[0, c]:
            continue
        nb = [n for n in neighbours(c) if s_len[n] > 0]
        m_c = sum(s_len[n] for n in nb)
        q_c = int(host[1, c])
        if not (m_c > 2 * 2 * params.k_for_knn and q_c > 0):
            continue
        valid_cell[c] = True
        for lo in range(0, q_c, chunk):
            jobs.append((c, min(chunk, q_c - lo), m_c))
            seg_src += [s_start[n] for n in nb]
            seg_len += [s_len[n] for n in nb]
    X_parts, H_parts, O_parts = [], [], []
    if jobs:
        J = len(jobs)
        T = sum(q for _, q, _ in jobs)
        tot = sum(seg_len)
        # ---- selected proxy points grouped by cell (stable: ascending index inside a cell)
        key = torch.where(sel & ops.h2d(valid_cell, torch.bool, device)[stored_cell.clamp(min=0)], stored_cell, big)
        rows = torch.sort(key, stable=True).indices[:T]
        X_sel = ps.proxy_points[rows]
        # ---- every job's surface cloud in one gather
        ints = ops.h2d(torch.tensor([seg_src, seg_len, [c for c, _, _ in jobs] + [0] * (len(seg_src) - J),
                                     [q for _, q, _ in jobs] + [0] * (len(seg_src) - J), [m for _, _, m in jobs] + [0] * (len(seg_src) - J)],
                                    dtype=torch.int64), torch.int64, device)
        src, ln = ints[0], ints[1]
        dst = torch.cumsum(ln, 0) - ln
        gather = torch.arange(tot, device=device) + torch.repeat_interleave(src - dst, ln, output_size=tot)
        S_all = torch.cat([ss.cells[k].cell_pts for k in s_keys if ss.cells[k].cell_pts.shape[0] > 0], dim=0)
        pc_all = S_all[gather].contiguous()
        job_cell, job_q, job_m = ints[2, :J], ints[3, :J], ints[4, :J]
        jid = torch.arange(J, device=device)
        cloud_of = torch.repeat_interleave(jid, job_m, output_size=tot).to(torch.int32)
        row_job = torch.repeat_interleave(jid, job_q, output_size=T).to(torch.int32)
        # ---- prediction boxes: cell centres in view space, 1 / (neighbourhood size x cell diagonal)   (:1468-1478)
        if Mv_host is not None:
            # J <= a few dozen jobs: their prediction boxes on the HOST (same fp32 arithmetic: torch CPU), one upload -- six small launches less
            jc = [c for c, _, _ in jobs]
            cw, dg = tab["centers_host"][jc], tab["diag_host"][jc]
            cen_h = (torch.cat((cw, torch.ones(J, 1)), 1) @ Mv_host.reshape(4, 4))[:, :3]
            inv_h = (1.0 / (params.prediction_neighborhood_size * dg)).float()
            up = ops.h2d(torch.cat((cen_h.reshape(-1), inv_h, Mv_host.reshape(1, 16).expand(J, -1).reshape(-1))), torch.float32, device)
            centers, inv_diag, MvJ = up[:3 * J].view(J, 3), up[3 * J:4 * J], up[4 * J:].view(J, 16)
        else:
            centers_w, diag = tab["centers"][job_cell], tab["diag"][job_cell]               # per job, gathered from the per-scene tables
            centers = (torch.cat((centers_w, torch.ones(J, 1, device=device)), 1) @ Mv)[:, :3].contiguous()
            inv_diag = (1.0 / (params.prediction_neighborhood_size * diag)).float().contiguous()
            MvJ = Mv.reshape(1, 16).expand(J, -1).contiguous()
        ops.transform_points_batched_(pc_all, MvJ, centers, inv_diag, cloud_of=cloud_of)
        X_q = ops.transform_points_batched_(X_sel.clone().contiguous(), MvJ, centers, inv_diag, cloud_of=row_job)
        # ---- view states -> prediction frame -> harmonics, all rows at once   (:1486-1497)
        vs = ps.view_states[rows].view(1, T, params.n_view_state_cameras)
        vs = su.move_view_state_to_view_space(vs, ((Mv_host if Mv_host is not None else Mv)[:3, :3].contiguous()
                                                   if torch.is_tensor(prediction_camera) else prediction_camera),
                                              n_elev=params.view_state_n_elev, n_azim=params.view_state_n_azim)
        base_harmonics, h_polar, h_azim = su.get_all_harmonics_under_degree(params.harmonic_degree, params.view_state_n_elev,
                                                                             params.view_state_n_azim, device)
        vh = su.compute_view_harmonics(vs, base_harmonics, h_polar, h_azim, params.view_state_n_elev, params.view_state_n_azim)[0]
        # ---- occupancy of all jobs
        if use_supervision_occ_instead_of_predicted:
            occ = ps.proxy_supervision_occ[rows]
        else:
            occ_net = getattr(macarons, "occupancy", macarons)
            def ragged(pc_, sizes_m_, X_, vh_, sizes_q_, draws):
                if isinstance(draws, dict):                 # index arrays drawn on the device (or handed back by a caller)
                    return occ_net.forward_ragged(pc_, sizes_m_, X_, vh_, sizes_q_, index_arrays=draws).view(-1, 1)
                return occ_net.forward_ragged(pc_, sizes_m_, X_, vh_, sizes_q_, perms=draws, perm_source=perm_source).view(-1, 1)

            if hasattr(occ_net, "forward_ragged") and world > 1:
                sizes_m, sizes_q = [m for _, _, m in jobs], [q for _, q, _ in jobs]
                if ragged_perms is None:                # rank 0 draws for every job, in job order (what the 1-rank pass draws)
                    ragged_perms = (_broadcast_job_index_arrays(occ_net, sizes_m, device, group, rank) if perm_source == "device"
                                    else _broadcast_job_perms(occ_net, sizes_m, device, group, rank))
                t0, t1 = mdist.shard_range(T, rank, world)
                q_start = np.concatenate(([0], np.cumsum(sizes_q)))
                m_start = np.concatenate(([0], np.cumsum(sizes_m)))
                mine = [j for j in range(J) if q_start[j] < t1 and q_start[j + 1] > t0]
                if mine:
                    q_l = [int(min(q_start[j + 1], t1) - max(q_start[j], t0)) for j in mine]
                    p0, p1 = int(m_start[mine[0]]), int(m_start[mine[-1] + 1])
                    draws_l = (_slice_index_arrays(occ_net, ragged_perms, sizes_m, mine[0], mine[-1] + 1) if isinstance(ragged_perms, dict)
                               else [ragged_perms[j] for j in mine])
                    occ_l = ragged(pc_all[p0:p1].contiguous(), [sizes_m[j] for j in mine], X_q[t0:t1].contiguous(),
                                   vh[t0:t1].contiguous(), q_l, draws_l)
                else:                                   # empty row shard (T < world): no kernels, the all-gather is joined
                    occ_l = torch.zeros(0, 1, dtype=torch.float32, device=device)
                occ = mdist.allgather_rows(occ_l, T, group)
            elif hasattr(occ_net, "forward_ragged"):
                occ = ragged(pc_all, [m for _, _, m in jobs], X_q, vh, [q for _, q, _ in jobs], ragged_perms)
                ragged_perms = occ_net.last_ragged_perms
            else:                                       # any other module with the reference's call signature: job by job
                outs, r0, p0 = [], 0, 0
                for _, q, m in jobs:
                    outs.append(macarons(mode='occupancy', partial_point_cloud=pc_all[p0:p0 + m][None], proxy_points=X_q[r0:r0 + q][None],
                                         view_harmonics=vh[r0:r0 + q][None]).view(-1, 1))
                    r0, p0 = r0 + q, p0 + m
                occ = torch.cat(outs)
            if record is not None:
                record["ragged_perms"] = ragged_perms
        ps.proxy_proba[rows] = occ                                                                # :1525
        X_parts, H_parts, O_parts = [X_sel], [vh], [occ]
    # indices of the never-seen points in ascending order, n_oof known: exclusive ranks scattered into place (no read-back)
    pos = torch.cumsum(oof_mask, 0) - 1
    oof_idx = torch.zeros(n_oof + 1, dtype=torch.int64, device=device).scatter_(
        0, torch.where(oof_mask, pos, torch.full_like(pos, n_oof)), torch.arange(P, device=device))[:n_oof]
    oof_X = ps.proxy_points[oof_idx]
    X_world = torch.cat(X_parts + [oof_X])
    view_harmonics = torch.cat(H_parts + [torch.zeros(len(oof_X), nh, device=device)])
    occ_probs = torch.cat(O_parts + [ps.proxy_proba[oof_idx]])
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


def _broadcast_job_index_arrays(occ_net, cloud_sizes, device, group, rank):
    """perm_source="device" on several ranks: rank 0 draws the index arrays of ALL jobs on its device; one broadcast of
    [g_idx | idx1 | idx2] (the offsets and lengths follow from the cloud sizes on every rank)."""
    from .. import dist as mdist
    J, Lg = len(cloud_sizes), occ_net.seq_len
    sz = [occ_net.scale_sizes(int(m_)) for m_ in cloud_sizes]
    n1, n2 = sum(s_[1] for s_ in sz), sum(s_[2] for s_ in sz)
    if rank == 0:
        ia = occ_net.ragged_index_arrays_device(cloud_sizes, device)
        buf = torch.cat((ia["g_idx"], ia["idx1"], ia["idx2"]))
    else:
        buf = torch.empty(J * Lg + n1 + n2, dtype=torch.int64, device=device)
    mdist.broadcast(buf, 0, group)
    cum = lambda v: np.concatenate(([0], np.cumsum(v))).astype(np.int64)
    offs = ops.h2d(np.concatenate([cum([s_[1] for s_ in sz]), cum([s_[2] for s_ in sz]), np.asarray([min(s_[0], Lg) for s_ in sz], np.int64)]),
                   torch.int64, device)
    return {"g_idx": buf[:J * Lg], "idx1": buf[J * Lg:J * Lg + n1], "idx2": buf[J * Lg + n1:], "off1": offs[:J + 1],
            "off2": offs[J + 1:2 * J + 2], "g_len": offs[2 * J + 2:].to(torch.int32)}


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
