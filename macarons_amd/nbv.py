"""One SCONE next-best-view decision on the MI355X: the sequence macarons/testers/shapenet.py:126-172 runs per
step, as one host routine over the HIP kernels, optionally sharded over the GPUs of a node.

    view state + view harmonics (Q proxy points)  ->  SconeOcc (Q queries vs M surface points)
    -> occupancy-weighted sampling of seq_len proxy points -> SconeVis -> SH coverage gains over C cameras -> argmax

Multi-GPU: the occupancy pass shards the Q queries across ranks (the surface cloud is replicated) and
all-gathers the occupancies; sampling and SconeVis (cheap, N <= 2048) run redundantly with identical draws; the
C candidate cameras are block-partitioned and the only other exchange is the all-gather of each rank's
(best gain, global camera index).
"""
import os

import torch

from . import dist as mdist
from .utility import scone_utils as su


class ViewStateGrid:
    """The constant tables of the view-state lattice (get_all_harmonics_under_degree, scone_utils.py:714-738)."""

    def __init__(self, device, degree=8, n_elev=7, n_azim=14):
        self.n_elev, self.n_azim = n_elev, n_azim
        self.base_harmonics, self.h_polar, self.h_azim = su.get_all_harmonics_under_degree(degree, n_elev, n_azim, device)


_side_streams = {}


def _harmonics_beside(scone_occ, pc, Xl, harmonics_of):
    """(handle of scone_occ.forward_begin(pc, Xl), harmonics_of(Xl)).  The first long kernel of a decision (neighbour search + local
    transformer over the whole cloud) needs neither the harmonics nor the down-sampled clouds: it is queued first, so that the host
    work of everything after it hides behind it, and the view-state / harmonics kernels run on a side stream beside it instead of
    in front of the rest.  Falls back to the plain order when the split forward does not apply."""
    dev = Xl.device
    main = torch.cuda.current_stream(dev)
    fork = torch.cuda.Event()
    fork.record(main)                                   # the inputs are ready here; what is queued next need not be waited for
    begun = scone_occ.forward_begin(pc, Xl)
    if begun is None or os.environ.get("MCR_NBV_SIDE") == "0":      # (A/B only: harmonics on the caller's stream after phase 1)
        return begun, harmonics_of(Xl)
    side = _side_streams.get((dev.index, main.cuda_stream))
    if side is None:
        side = _side_streams[(dev.index, main.cuda_stream)] = torch.cuda.Stream(device=dev)
    side.wait_event(fork)
    with torch.cuda.stream(side):
        vh = harmonics_of(Xl)
    main.wait_stream(side)
    vh.record_stream(main)
    return begun, vh


def _guarded(impl, scone_occ, range_guard, group, draws, device, scone_vis=None):
    """Run one decision with SconeOcc's range check deferred to the END of the step (the step itself stays free of host
    synchronisation); if the flag comes back set (an activation left the fp16 range of the default matrix path, SconeOcc.range_guard)
    the decision is repeated on variant 5 with the SAME hidden draws.  `draws()` pins the draws before the first attempt when
    the caller did not.  With several ranks the flag is all-reduced first: every rank repeats, or none."""
    from . import ops
    capturing = torch.cuda.is_current_stream_capturing()
    prev = scone_occ.range_guard
    guard = range_guard and prev != "off" and ops.current_variant() in (6, 7)
    scone_occ.range_guard = "defer" if (guard or prev != "off") else "off"
    # SconeVis (its encoders run on the same fp16 planes) reports into the SAME flag: one read-back covers both networks
    vis_prev = None
    if scone_vis is not None and hasattr(scone_vis, "range_guard"):
        vis_prev = (scone_vis.range_guard, scone_vis._range_flag)
        scone_vis.range_guard = scone_occ.range_guard
    try:
        # the flag exists on every rank before the step (not only on ranks that run a guarded forward: a rank with an empty query
        # shard runs none), so that whether the all-reduce below is entered depends on rank-invariant state only
        scone_occ.clear_range_flag(device if guard else None)
        if vis_prev is not None:
            scone_vis._range_flag = scone_occ.range_flag()
        kw = draws() if (guard and not capturing) else {}
        out = impl(**kw)
        flag = scone_occ.range_flag() if ops.current_variant() in (6, 7) else None
        out["range_flag"] = flag
        if guard and not capturing:
            xch = mdist.exchange_on(group)
            if xch:
                flag = mdist.all_reduce_max(flag, group)
            # the one read-back, after the last kernel of the decision was queued; the decision itself rides along (a caller that
            # wants the camera index on the host reads out["host"]["nbv_idx"] instead of paying a second device->host round trip)
            if not xch and "record" in out:          # written by the decision's last kernel (mcr_nbv_decide): nothing to assemble
                both = out["record"].cpu()
            else:
                both = torch.cat((flag.to(out["nbv_idx"].device).view(-1).to(torch.float64), out["nbv_idx"].view(-1).to(torch.float64),
                                  out["max_gain"].view(-1).to(torch.float64))).cpu()
            n_dec = out["nbv_idx"].numel()
            out["host"] = {"nbv_idx": both[1:1 + n_dec].to(torch.int64), "max_gain": both[1 + n_dec:].to(torch.float32)}
            if int(both[0]):
                with ops.variant(5):                 # (scoped to this thread's calls: the process default is not touched)
                    out = impl(**kw)
                out["range_flag"], out["fallback_variant"] = None, 5
                out.pop("host", None)
    finally:
        scone_occ.range_guard = prev
        if vis_prev is not None:
            scone_vis.range_guard, scone_vis._range_flag = vis_prev
    return out


def nbv_step(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len=2048, min_occ=0.1,
             max_points_per_pass=300000, true_monte_carlo_sampling=True, occ_perms=None, samples=None, group=None,
             view_proj=None, filter_tol=0.01, return_samples=False, range_guard=True):
    """One decision; see _nbv_step for the arguments.  range_guard: check once, at the end of the step, whether an activation left
    the fp16 range of the default matrix path and repeat the decision on the full-range variant 5 if so (dict key
    `fallback_variant`); False: no read-back at all, the device flag is returned as `range_flag` for the caller to look at."""
    if X.shape[0] > 1:                              # a scene batch: nbv_step_batch (B independent decisions, SURVEY §8e sharding rule)
        if view_proj is not None or max_points_per_pass < X.shape[1] * X.shape[0]:
            raise NotImplementedError("nbv_step: the proxy filter / multi-chunk occupancy pass are single-cloud options")
        return nbv_step_batch(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len=seq_len, min_occ=min_occ,
                              true_monte_carlo_sampling=true_monte_carlo_sampling, occ_perms=occ_perms, samples=samples,
                              group=group, return_samples=return_samples, range_guard=range_guard)
    fixed = dict(occ_perms=occ_perms, samples=samples)

    def draws():                                    # a repeat must see the first attempt's draws: pin them up front (one chunk)
        if fixed["occ_perms"] is None and X.shape[1] <= max_points_per_pass and view_proj is None:
            fixed["occ_perms"] = scone_occ.draw_perms(pc.shape[1])
        if fixed["samples"] is None:
            fixed["samples"] = torch.rand(seq_len, 1, device=X.device)
        return {}

    def impl():
        return _nbv_step(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len, min_occ, max_points_per_pass,
                         true_monte_carlo_sampling, fixed["occ_perms"], fixed["samples"], group, view_proj, filter_tol, return_samples)
    if mdist.group_world_rank(group)[0] > 1:
        draws = lambda: {}                          # noqa: E731  (sharded: rank 0's draws are broadcast inside the step; a repeat redraws)
    return _guarded(impl, scone_occ, range_guard, group, draws, X.device, scone_vis)


def nbv_step_one_rank_of(world, scone_occ, scone_vis, pc, X, X_view, X_cam, grid, occ_perms, samples, rank=0, seq_len=2048, min_occ=0.1):
    """MEASUREMENT AID (bench.py: nbv_step.one_rank_of_8): what ONE rank of a `world`-rank sharded step computes for this decision, run
    on this GPU alone -- its 1/world of the queries through SconeOcc, then the part every rank repeats (sampling, SconeVis, the decision)
    and its 1/world of the cameras, with the two exchanges replaced by local copies of the same size.  The per-rank critical path apart
    from collective latency; the decision it returns is NOT the real one.  No read-back inside (range check deferred: range_flag)."""
    return _guarded(lambda: _nbv_step(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len, min_occ, occ_perms=occ_perms, samples=samples,
                                      _emulate=(rank, world)), scone_occ, False, None, lambda: {}, X.device, scone_vis)


def _nbv_step(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len=2048, min_occ=0.1,
              max_points_per_pass=300000, true_monte_carlo_sampling=True, occ_perms=None, samples=None, group=None,
              view_proj=None, filter_tol=0.01, return_samples=False, _emulate=None):
    """pc [1,M,3] surface points, X [1,Q,3] proxy points, X_view [n_view,3] past camera positions, X_cam [C,3]
    candidate cameras (all in the normalised prediction-view space, as the reference feeds its networks).
    Returns dict(gains [C_local or C], nbv_idx (global camera index), max_gain, occ [Q,1], n_unique) -- all device tensors
    (n_unique: int32 [1]); nothing is read back inside the step, so it runs without a single host synchronisation.
    `occ_perms` / `samples` pin the hidden RNG draws (SconeOcc randperms; sampling uniforms).  `view_proj` [n_view,4,4]
    (full-projection matrices of the past views) switches on the tester's proxy-point filter (testers/shapenet.py:117-122)."""
    world, rank = mdist.group_world_rank(group)            # group=None: local (pass torch.distributed.group.WORLD to shard)
    # the exchange path (broadcast of the hidden draws, all-gathers, record merge) normally needs world > 1; the env knob runs it
    # through RCCL on a single rank too, so that a one-GPU box can test it end to end
    sharded = mdist.exchange_on(group)
    gather_rows, gather_best = mdist.allgather_rows, mdist.allgather_best
    if _emulate is not None:
        # TIMING ONLY (bench.py's N = 1 line): what rank `r` of a `w`-rank job computes, on this one GPU -- its query shard, the
        # redundant part (sampling, SconeVis, its camera shard) -- with the two exchanges replaced by local stand-ins of the same
        # size (the other ranks' occupancies are copies of this rank's): the critical path of one rank, i.e. what a w-GPU step
        # costs apart from the collectives' latency.  The decision it returns is not the real one.
        rank, world = _emulate
        sharded = True
        gather_rows = lambda t, n, g_=None: t.repeat(-(-n // max(t.shape[0], 1)), 1)[:n].contiguous()      # noqa: E731
        def gather_best(g_, c0_, grp=None):                                                             # noqa: E306
            b_ = torch.max(g_, dim=1)
            return b_.values, b_.indices + c0_
    dev = X.device
    if view_proj is not None:                                                       # testers/shapenet.py:122
        X = su.filter_proxy_points(view_proj, X[0], pc.reshape(-1, 3), filter_tol=filter_tol)[0][None]
    Q = X.shape[1]
    if Q == 0:
        raise ValueError("nbv_step: no proxy point left to score (the proxy filter removed every point)")
    with torch.no_grad():
        # ---- view state -> harmonics for this rank's slice of the queries (testers/shapenet.py:126-130) ----
        q0, q1 = mdist.shard_range(Q, rank, world)
        Xl = X[:, q0:q1].contiguous()
        if sharded and _emulate is None and (occ_perms is None or samples is None):
            # one chunk per rank; the hidden draws (SconeOcc.forward's three randperms from the CPU generator, the sampling
            # uniforms from the device generator) are rank 0's and reach the others in ONE broadcast (the uniforms travel
            # bit-cast inside the int64 buffer): every rank drawing its own would silently diverge
            draw_perms, draw_u = occ_perms is None, samples is None
            if draw_perms:
                occ_perms = [p.to(dev) for p in scone_occ.draw_perms(pc.shape[1])]
            if draw_u:
                samples = torch.rand(seq_len, 1, device=dev)
            got_p, got_u = mdist.broadcast_draws(occ_perms if draw_perms else [], samples if draw_u else None, 0, group)
            if draw_perms:
                occ_perms = got_p
            if draw_u:
                samples = got_u
        if q1 > q0:
            harmonics_of = lambda pts: su.compute_view_harmonics(su.compute_view_state(pts, X_view, grid.n_elev, grid.n_azim),   # noqa: E731
                                                                 grid.base_harmonics, grid.h_polar, grid.h_azim, grid.n_elev, grid.n_azim)
            begun, vh_l = _harmonics_beside(scone_occ, pc, Xl, harmonics_of) if occ_perms is not None else (None, harmonics_of(Xl))
            # ---- occupancy (:139-144) ----
            if occ_perms is not None:
                occ_l = scone_occ(pc, Xl, vh_l, perms=occ_perms, begun=begun).view(-1, 1)
            else:
                occ_l = su.compute_occupancy_probability(scone_occ, pc, Xl, vh_l, max_points_per_pass=max_points_per_pass).view(-1, 1)
        else:                                       # empty query shard (Q < world): no kernels, but the collectives below are joined
            vh_l = torch.zeros(1, 0, grid.base_harmonics.shape[0], dtype=torch.float32, device=dev)
            occ_l = torch.zeros(0, 1, dtype=torch.float32, device=dev)
        if sharded:
            # only the occupancies travel (4 B per proxy point); the view harmonics of the <= seq_len sampled points are
            # recomputed locally below (a row is a function of its own point): all-gathering them would move 256 B per point
            occ = gather_rows(occ_l, Q, group)
            vh = None
        else:
            occ, vh = occ_l, vh_l[0]
        # ---- occupancy-weighted Monte-Carlo sampling (:146-154); identical on every rank ----
        if samples is None:                          # (the sharded path drew and broadcast them above)
            samples = torch.rand(seq_len, 1, device=dev)
        # no host read-back: the unique sampled points stay padded to seq_len rows and their count stays on the device
        # (SconeVis consumes it as `lengths`); the reference slices on the host (:146-157)
        if vh is not None:
            proxy_points, vh_s, sample_idx, n_unique = su.sample_proxy_points(X[0], occ, vh, n_sample=seq_len, min_occ=min_occ,
                                                                              samples=samples, padded=True)
        else:
            from . import ops
            proxy_points, _, sample_idx, _, n_unique = ops.sample_proxy(X[0].contiguous(), occ.reshape(-1), None, samples.reshape(-1),
                                                                        min_occ, padded=True)
            vs_s = su.compute_view_state(proxy_points[None, :, :3].contiguous(), X_view, grid.n_elev, grid.n_azim)
            vh_s = su.compute_view_harmonics(vs_s, grid.base_harmonics, grid.h_polar, grid.h_azim, grid.n_elev, grid.n_azim)[0]
        sampled = (proxy_points, sample_idx)
        # ---- visibility-gain harmonics (:157-160) ----
        harm = scone_vis(proxy_points[None], view_harmonics=vh_s[None], lengths=n_unique)
        if true_monte_carlo_sampling:
            proxy_points, harm = proxy_points[sample_idx][None].contiguous(), harm[0][sample_idx][None].contiguous()
        else:                                       # score the unique points once each: their count is needed on the host
            k = int(n_unique)
            proxy_points, harm = proxy_points[:k][None].contiguous(), harm[:, :k].contiguous()
        # ---- coverage gains over this rank's camera shard + arg-max (:167-172) ----
        C = X_cam.shape[0]
        c0, c1 = mdist.shard_range(C, rank, world)
        if c1 > c0:
            gains = scone_vis.compute_coverage_gain(proxy_points, harm, X_cam[c0:c1].contiguous().view(1, -1, 3))
        else:                                       # empty camera shard (C < world)
            gains = torch.zeros(1, 0, dtype=torch.float32, device=dev)
        # no proxy point above min_occ: nothing was sampled (the kernels then score one zero row): the reference fails on the empty
        # sample (scone_utils.py:1052-1061); here the decision says so on the device: gains NaN, nbv_idx -1
        record = None
        if sharded:
            empty = n_unique.view(1, 1) < 1
            gains = torch.where(empty, torch.full_like(gains, float("nan")), gains)
            max_gain, nbv_idx = gather_best(gains, c0, group)
            nbv_idx = torch.where(empty.view(-1), torch.full_like(nbv_idx, -1), nbv_idx)
        else:                                       # the same rules + the host's read-back record in one launch (mcr_nbv_decide)
            from . import ops
            flag = scone_occ.range_flag() if scone_occ.range_guard != "off" else None
            max_gain, nbv_idx, record = ops.nbv_decide(gains, n_unique.view(-1), flag)
    out = {"gains": gains[0], "cam_range": (c0, c1), "nbv_idx": nbv_idx, "max_gain": max_gain, "occ": occ,
           "n_unique": n_unique}
    if record is not None:
        out["record"] = record
    if return_samples:                              # the unique sampled proxy points (first n_unique of seq_len rows) and the inverse map [seq_len]
        out["proxy_points"], out["sample_idx"] = sampled
    return out


def draw_batch(scone_occ, B, M, seq_len, device):
    """The hidden draws of B single-cloud decisions made one after the other (testers/shapenet.py:33-37 loops the objects of a
    batch): per cloud SconeOcc's three randperms from the CPU generator, then its seq_len sampling uniforms from the device
    generator.  -> (perms: three int64 tensors [B, n_i], samples [B, seq_len])."""
    per_cloud = [scone_occ.draw_perms(M) for _ in range(B)]
    perms = [torch.stack([pc_[i] for pc_ in per_cloud]).to(device) for i in range(3)]
    samples = torch.stack([torch.rand(seq_len, 1, device=device).view(-1) for _ in range(B)])
    return perms, samples


def nbv_step_batch(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len=2048, min_occ=0.1, true_monte_carlo_sampling=True,
                   occ_perms=None, samples=None, group=None, return_samples=False, range_guard=True):
    """B independent decisions in one launch sequence; see _nbv_step_batch.  range_guard as in nbv_step."""
    fixed = dict(occ_perms=occ_perms, samples=samples)
    xch = mdist.exchange_on(group)

    def draws():
        if not xch and (fixed["occ_perms"] is None or fixed["samples"] is None):
            dp, du = draw_batch(scone_occ, X.shape[0], pc.shape[1], seq_len, X.device)
            fixed["occ_perms"] = dp if fixed["occ_perms"] is None else fixed["occ_perms"]
            fixed["samples"] = du if fixed["samples"] is None else fixed["samples"]
        return {}

    def impl():
        return _nbv_step_batch(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len, min_occ, true_monte_carlo_sampling,
                               fixed["occ_perms"], fixed["samples"], group, return_samples)
    return _guarded(impl, scone_occ, range_guard, group, draws, X.device, scone_vis)


def _nbv_step_batch(scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len=2048, min_occ=0.1, true_monte_carlo_sampling=True,
                    occ_perms=None, samples=None, group=None, return_samples=False):
    """B independent NBV decisions (a scene batch: BASELINE config 3 = 8 objects x 32k proxy points x 200 cameras) in ONE launch
    sequence.  pc [B,M,3], X [B,Q,3], X_view [B,n_view,3] (every object has its own trajectory) or [n_view,3], X_cam [C,3] or
    [B,C,3].  occ_perms: three int64 tensors [B, n_i] (or 1-D, shared), samples [B, seq_len]; None = draw_batch().  Cloud b's
    result equals nbv_step(pc[b:b+1], X[b:b+1], X_view[b], X_cam[b], occ_perms=[p[b] ...], samples=samples[b]).

    Sharding over the ranks of `group` (block-partition the flattened scene-batch x camera product):
      B >= world: the CLOUDS are sharded -- every rank runs whole decisions for its clouds (nothing replicated, no data-path
                  collective) and the only exchange is the all-gather of one 8-byte (gain, camera) record per cloud;
      B <  world: clouds replicated, the Q queries of the occupancy pass and the C cameras are sharded as in nbv_step
                  (all-gather of the occupancies, then of the per-rank records).
    Returns dict(gains [B_local, C_local], cloud_range, cam_range, max_gain [B], nbv_idx [B] int64, occ [B_local, Q, 1],
    n_unique int32 [B_local]); device tensors, no host synchronisation inside."""
    from . import ops
    world, rank = mdist.group_world_rank(group)            # group=None: local (pass torch.distributed.group.WORLD to shard)
    dev = X.device
    B, Q, M, C = X.shape[0], X.shape[1], pc.shape[1], X_cam.shape[-2]
    if pc.shape[0] != B:
        raise ValueError("nbv_step_batch: pc and X must hold the same number of clouds")
    xch = mdist.exchange_on(group)                         # (a one-rank group with MCR_FORCE_DIST_PATH runs every collective too)
    by_cloud = xch and B >= world
    with torch.no_grad():
        # ---- hidden draws: rank 0's, for ALL clouds, in one broadcast ----
        if occ_perms is None or samples is None:
            dp, du = draw_batch(scone_occ, B, M, seq_len, dev)
            if xch:
                got_p, got_u = mdist.broadcast_draws(dp if occ_perms is None else [], du if samples is None else None, 0, group)
                dp, du = (got_p if occ_perms is None else dp), (got_u if samples is None else du)
            occ_perms = dp if occ_perms is None else occ_perms
            samples = du if samples is None else samples
        samples = samples.reshape(B, seq_len).to(dev)
        occ_perms = [p.to(dev) for p in occ_perms]
        b0, b1 = mdist.shard_range(B, rank, world) if by_cloud else (0, B)
        q0, q1 = (0, Q) if (by_cloud or not xch) else mdist.shard_range(Q, rank, world)
        c0, c1 = (0, C) if (by_cloud or not xch) else mdist.shard_range(C, rank, world)
        Bl = b1 - b0
        pc_l, X_b = pc[b0:b1].contiguous(), X[b0:b1].contiguous()
        Xv_l = X_view[b0:b1].contiguous() if X_view.dim() == 3 else X_view
        perms_l = [p[b0:b1] if p.dim() == 2 else p for p in occ_perms]
        u_l = samples[b0:b1].contiguous()
        vs_of = lambda pts: su.compute_view_harmonics(su.compute_view_state(pts, Xv_l, grid.n_elev, grid.n_azim), grid.base_harmonics,
                                                      grid.h_polar, grid.h_azim, grid.n_elev, grid.n_azim)
        # ---- view harmonics + occupancy of this rank's (clouds, queries) block ----
        if q1 > q0:
            Xl = X_b[:, q0:q1].contiguous()
            begun, vh_l = _harmonics_beside(scone_occ, pc_l, Xl, vs_of)
            occ_l = scone_occ(pc_l, Xl, vh_l, perms=perms_l, begun=begun).view(Bl, q1 - q0)
        else:
            vh_l, occ_l = None, torch.zeros(Bl, 0, dtype=torch.float32, device=dev)
        if q1 - q0 < Q:                               # query-sharded: only the occupancies travel (4 B per proxy point and cloud)
            occ = mdist.allgather_rows(occ_l.t().contiguous(), Q, group).t().contiguous()
            vh_l = None
        else:
            occ = occ_l
        # ---- sampling (per cloud, own uniforms), SconeVis on the padded unique sets ----
        res, resh, inv, uniq, nu, _ = ops.sample_proxy_batched(X_b, occ, vh_l, u_l, min_occ)
        if resh is None:                              # harmonics of the sampled points recomputed locally (a row depends on its point only)
            resh = vs_of(res[..., :3].contiguous())
        harm = scone_vis(res, view_harmonics=resh, lengths=nu)
        if true_monte_carlo_sampling:
            pts_s = torch.gather(res, 1, inv[..., None].expand(-1, -1, 4))
            harm_s = torch.gather(harm, 1, inv[..., None].expand(-1, -1, 64))
        else:
            raise NotImplementedError("nbv_step_batch scores the Monte-Carlo multiset (true_monte_carlo_sampling=True)")
        # ---- gains over this rank's cameras, arg-max exchange ----
        cams = X_cam[b0:b1] if X_cam.dim() == 3 else X_cam[None].expand(Bl, -1, -1)
        cams = cams[:, c0:c1].contiguous()
        gains = scone_vis.compute_coverage_gain(pts_s, harm_s, cams) if c1 > c0 else torch.zeros(Bl, 0, dtype=torch.float32, device=dev)
        record = None
        if not xch:                                   # NaN rule + arg-max + the host's read-back record in one launch (mcr_nbv_decide)
            flag = scone_occ.range_flag() if scone_occ.range_guard != "off" else None
            max_gain, nbv_idx, record = ops.nbv_decide(gains, nu.view(-1), flag)
        else:
            gains = torch.where(nu.view(-1, 1) < 1, torch.full_like(gains, float("nan")), gains)  # nothing sampled: NaN gains, index -1
            if by_cloud:
                rec = mdist.allgather_rows(ops.best_record(gains, 0), B, group)
                max_gain, nbv_idx = rec[:, 0].contiguous(), rec[:, 1].to(torch.int64)
                # a rank only knows its own clouds' counts; the NaN max_gain marks the others
                nbv_idx = torch.where(torch.isnan(max_gain), torch.full_like(nbv_idx, -1), nbv_idx)
            else:
                max_gain, nbv_idx = mdist.allgather_best(gains, c0, group)
                nbv_idx = torch.where(nu < 1, torch.full_like(nbv_idx, -1), nbv_idx)
    out = {"gains": gains, "cloud_range": (b0, b1), "cam_range": (c0, c1), "max_gain": max_gain, "nbv_idx": nbv_idx,
           "occ": occ.view(Bl, Q, 1), "n_unique": nu}
    if record is not None:
        out["record"] = record
    if return_samples:
        out["proxy_points"], out["sample_idx"] = res, inv
    return out


class GraphedNbvStep:
    """One NBV decision captured in a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~60 kernel launches of the sync-free
    `nbv_step` cost one graph launch, which removes the launch gaps between the many small kernels (single-GPU path only: the
    RCCL exchange is not captured).

    The step's host-side randomness cannot live inside a graph, so it becomes an input: every call draws SconeOcc's three
    `torch.randperm` (CPU generator, the reference's order) and the `seq_len` sampling uniforms (device generator) exactly as
    the eager step would, copies them and the scene tensors into the static input buffers, and replays.  Shapes are fixed at
    construction.  Returns the same dict as `nbv_step` (static output tensors: clone what must survive the next call)."""

    def __init__(self, scone_occ, scone_vis, pc, X, X_view, X_cam, grid, seq_len=2048, min_occ=0.1, warmup=2, one_rank_of=None):
        """one_rank_of (measurement aid): capture nbv_step_one_rank_of(one_rank_of, ...) -- rank 0's share of a sharded step -- instead
        of the whole step."""
        self.one_rank_of = one_rank_of
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError("GraphedNbvStep captures the single-GPU step; use nbv_step under torchrun")
        self.occ, self.vis, self.grid, self.seq_len, self.min_occ = scone_occ, scone_vis, grid, seq_len, min_occ
        dev = X.device
        self._in = {"pc": pc.clone(), "X": X.clone(), "X_view": X_view.clone(), "X_cam": X_cam.clone(),
                    "samples": torch.empty(seq_len, 1, device=dev),
                    "perms": [p.to(dev) for p in scone_occ.draw_perms(pc.shape[1])]}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                  # warm-up on the capture stream: weight tables, blobs and arenas get built
            for _ in range(max(1, warmup)):
                self._run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._out = self._run()
        torch.cuda.synchronize(dev)

    def _run(self):
        i = self._in
        if self.one_rank_of:
            return nbv_step_one_rank_of(self.one_rank_of, self.occ, self.vis, i["pc"], i["X"], i["X_view"], i["X_cam"], self.grid, i["perms"],
                                        i["samples"], seq_len=self.seq_len, min_occ=self.min_occ)
        return nbv_step(self.occ, self.vis, i["pc"], i["X"], i["X_view"], i["X_cam"], self.grid, seq_len=self.seq_len,
                        min_occ=self.min_occ, occ_perms=i["perms"], samples=i["samples"], return_samples=True, range_guard=False)

    def __call__(self, pc=None, X=None, X_view=None, X_cam=None, occ_perms=None, samples=None):
        i = self._in
        for name, t in (("pc", pc), ("X", X), ("X_view", X_view), ("X_cam", X_cam)):
            if t is not None:
                i[name].copy_(t)
        perms = occ_perms if occ_perms is not None else self.occ.draw_perms(i["pc"].shape[1])
        for dst, src in zip(i["perms"], perms):
            dst.copy_(src, non_blocking=True)
        if samples is not None:
            i["samples"].copy_(samples.reshape(self.seq_len, 1))
        else:
            i["samples"].uniform_()                     # == torch.rand(seq_len, 1, device=dev) of the eager step
        self.graph.replay()
        return self._out
