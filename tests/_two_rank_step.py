"""Helper of tests/test_nbv_gpu.py: the sharded NBV step on TWO ranks (launched with torch.distributed.run, 2 processes).

MCR_TEST_BACKEND = nccl (two GPUs, RCCL) or gloo (both ranks on cuda:0 of a one-GPU box: RCCL refuses duplicate devices, so the
collectives run host-staged -- macarons_amd/dist.py -- while every kernel still runs on the GPU).  Every rank first computes the
1-rank answers (no process group yet), then the 2-rank ones, and compares bit for bit."""
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


def batch_scene(g, B, dev):
    """B different grid clouds out of one golden: cloud b has its axes rotated cyclically b times (stays on the 2^-10 grid), its
    own past views, hidden draws and uniforms."""
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    roll = lambda a, b: np.roll(a, b, axis=-1)
    pc = T(np.stack([roll(g["pc"][0], b) for b in range(B)]))
    X = T(np.stack([roll(g["X"][0], b) for b in range(B)]))
    cams = g["X_cam"]
    X_view = T(np.stack([cams[[b % len(cams), (3 * b + 1) % len(cams)]] for b in range(B)]))
    gen = torch.Generator().manual_seed(77)
    M = pc.shape[1]
    perms = [torch.stack([torch.randperm(M, generator=gen)[:2048] for _ in range(B)])]
    ds = max(2, int(np.power(M / 128, 0.5)))
    m = M
    for _ in range(2):
        perms.append(torch.stack([torch.randperm(m, generator=gen)[:m // ds] for _ in range(B)]))
        m //= ds
    u = torch.rand(B, 2048, generator=gen).to(dev)
    return pc, X, X_view, T(cams), perms, u


def macarons_decisions(dev, group=None):
    """The two consecutive MACARONS decisions of the reference golden (tests/golden/macarons_decision.npz) on fresh scenes; with
    `group` the decision is sharded over its ranks.  -> per decision (result dict, snapshot of the proxy-scene state)."""
    from types import SimpleNamespace as NS
    import test_macarons_regime_gpu as tm
    from macarons_amd.utility import macarons_utils as mu
    g = golden("macarons_decision")
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    m = tm._models(dev)
    surface, proxy = tm._decision_scenes(g, dev)
    H, W = int(g["hw"][0]), int(g["hw"][1])
    params = NS(n_harmonics=64, harmonic_degree=8, view_state_n_elev=7, view_state_n_azim=14, k_for_knn=16,
                prediction_neighborhood_size=3, n_view_state_cameras=98, sensor_range=40., min_occ_for_proxy_points=0.1, seq_len=2048,
                distance_factor_th=17., image_height=H, image_width=W, carving_tolerance=0.05)
    dmask = np.unpackbits(g["dmask"])[:2 * H * W].reshape(2, H, W).astype(bool)
    out = []
    for c in range(2):
        cam = mu.SceneCamera(mu.camera_record(g["Mview"][c], g["Mfull"][c], g["ndc"], g["eyes"][c], params.sensor_range).to(dev),
                             T(g["eyes"][c:c + 1]), float(g["zfar"]))
        nrec = torch.stack([mu.camera_record(g[f"nMview_{c}"][k], g[f"nMfull_{c}"][k], g["ndc"], g["n_eyes"][c, k], params.sensor_range)
                            for k in range(5)]).to(dev)
        torch.manual_seed(5100 + c)
        with torch.no_grad():
            r = mu.macarons_nbv_decision(params, m, proxy, surface, cam, T(g["depth"][c]), T(dmask[c]), nrec, T(g["n_eyes"][c]), dev,
                                         samples=T(g[f"u_{c}"]), group=group)
        state = {k: getattr(proxy, k).clone() for k in ("view_states", "proxy_supervision_occ", "out_of_field", "proxy_n_inside_fov",
                                                         "proxy_n_behind_depth", "proxy_proba")}
        state["cells"] = torch.cat([proxy.cells[k].cell_features[:, 0] for k in sorted(proxy.cells)])
        out.append((r, state))
    return out, g


def main():
    rank, lr = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("MCR_TEST_BACKEND", "nccl")
    dev = torch.device("cuda", lr if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    from macarons_amd.networks import SconeVis, SconeOcc
    from macarons_amd.nbv import nbv_step, nbv_step_batch, ViewStateGrid
    occ, vis = SconeOcc(), SconeVis()
    sdo = weights.make_state_dict(weights.shapes_of(occ), 2)
    sdv = weights.make_state_dict(weights.shapes_of(vis), 1)
    sdo["linear3.bias"] = sdo["linear3.bias"] + np.float32(0.5)
    occ.load_state_dict({k: torch.from_numpy(v) for k, v in sdo.items()})
    vis.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()})
    occ, vis = occ.to(dev).eval(), vis.to(dev).eval()
    grid = ViewStateGrid(dev)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    g2, g1 = golden("e2e_grid_config2"), golden("e2e_grid_config1")
    P = lambda g: [torch.from_numpy(g[f"perm{i}"].astype(np.int64)) for i in range(3)]
    a2 = (occ, vis, T(g2["pc"]), T(g2["X"]), T(g2["X_view"]), T(g2["X_cam"]), grid)
    tiny = (occ, vis, T(g1["pc"]), T(g1["X"][:, :1]), T(g1["X_view"]), T(g1["X_cam"][:1]), grid)       # Q = 1 < world, C = 1 < world
    pcb, Xb, Xvb, camb, permb, ub = batch_scene(g1, 3, dev)
    ab = (occ, vis, pcb, Xb, Xvb, camb, grid)

    # ---- 1-rank answers (torch.distributed not initialised: the plain path) ----
    s_full = nbv_step(*a2, occ_perms=P(g2), samples=T(g2["samples"]))
    s_tiny = nbv_step(*tiny, occ_perms=P(g1), samples=T(g1["samples"]))
    s_b3 = nbv_step_batch(*ab, occ_perms=permb, samples=ub)
    s_b1 = nbv_step_batch(occ, vis, pcb[:1], Xb[:1], Xvb[:1], camb, grid, occ_perms=[p[:1] for p in permb], samples=ub[:1])
    mac1, gmac = macarons_decisions(dev)
    from macarons_amd import ops as _ops
    with _ops.variant(7):                                                # the opt-in 16-bit matrix path: 1-rank answers (case V below)
        s7_full = nbv_step(*a2, occ_perms=P(g2), samples=T(g2["samples"]))
        s7_b3 = nbv_step_batch(*ab, occ_perms=permb, samples=ub)
    torch.cuda.synchronize()

    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    fails = []

    def expect(cond, what):
        if not bool(cond):
            fails.append(what)

    # B0: the empty-shard case on a FRESH module (its range flag has never been created on rank 1, whose query shard is empty): every
    # rank must still enter the range-flag all-reduce (a rank that skipped it hung / mis-paired the collectives)
    occ_f = SconeOcc()
    occ_f.load_state_dict(occ.state_dict())
    occ_f = occ_f.to(dev).eval()
    r = nbv_step(occ_f, *tiny[1:], occ_perms=P(g1), samples=T(g1["samples"]), group=dist.group.WORLD)
    expect(torch.equal(r["occ"], s_tiny["occ"]) and torch.equal(r["max_gain"], s_tiny["max_gain"]) and int(r["nbv_idx"]) == 0, "B0 fresh module")

    # A: query- and camera-sharded single-cloud step (config-2 shape) == the 1-rank step, bit for bit; == the reference golden at 1e-4
    r = nbv_step(*a2, occ_perms=P(g2), samples=T(g2["samples"]), group=dist.group.WORLD)
    c0, c1 = r["cam_range"]
    expect((c0, c1) == ((0, 50) if rank == 0 else (50, 100)), "A cam_range")
    expect(torch.equal(r["occ"], s_full["occ"]), "A occ")
    expect(torch.equal(r["gains"], s_full["gains"][c0:c1]), "A gains")
    expect(torch.equal(r["max_gain"], s_full["max_gain"]) and int(r["nbv_idx"]) == int(s_full["nbv_idx"]) == int(g2["nbv_idx"]), "A decision")
    expect(float(np.abs(r["gains"].cpu().numpy() - g2["gains"][c0:c1]).max()) < 1e-4 * float(np.abs(g2["gains"]).max()), "A golden")
    # B: fewer queries and cameras than ranks: rank 1 holds empty shards but joins every collective
    r = nbv_step(*tiny, occ_perms=P(g1), samples=T(g1["samples"]), group=dist.group.WORLD)
    expect(r["cam_range"] == ((0, 1) if rank == 0 else (1, 1)), "B cam_range")
    expect(torch.equal(r["occ"], s_tiny["occ"]) and torch.equal(r["max_gain"], s_tiny["max_gain"]) and int(r["nbv_idx"]) == 0, "B decision")
    # C: scene batch, B = 3 >= world: clouds sharded (2 + 1), one record all-gather
    r = nbv_step_batch(*ab, occ_perms=permb, samples=ub, group=dist.group.WORLD)
    b0, b1 = r["cloud_range"]
    expect((b0, b1) == ((0, 2) if rank == 0 else (2, 3)) and r["cam_range"] == (0, camb.shape[0]), "C ranges")
    expect(torch.equal(r["occ"], s_b3["occ"][b0:b1]) and torch.equal(r["gains"], s_b3["gains"][b0:b1]), "C own clouds")
    expect(torch.equal(r["max_gain"], s_b3["max_gain"]) and torch.equal(r["nbv_idx"], s_b3["nbv_idx"]), "C decisions")
    # D: scene batch smaller than the world (B = 1 < 2): replicated cloud, queries and cameras sharded
    r = nbv_step_batch(occ, vis, pcb[:1], Xb[:1], Xvb[:1], camb, grid, occ_perms=[p[:1] for p in permb], samples=ub[:1],
                       group=dist.group.WORLD)
    c0, c1 = r["cam_range"]
    expect(r["cloud_range"] == (0, 1) and (c0, c1) == ((0, 10) if rank == 0 else (10, 20)), "D ranges")
    expect(torch.equal(r["occ"], s_b1["occ"]) and torch.equal(r["gains"], s_b1["gains"][:, c0:c1]), "D shards")
    expect(torch.equal(r["max_gain"], s_b1["max_gain"]) and torch.equal(r["nbv_idx"], s_b1["nbv_idx"]), "D decision")
    expect(torch.equal(s_b1["max_gain"], s_b3["max_gain"][:1]) and torch.equal(s_b1["occ"], s_b3["occ"][:1]), "D batch-of-1 == first of 3")
    # V: the same on VARIANT 7 (selected per call, on every rank's calling thread): the sharded step / the cloud-sharded batch reproduce
    # the 1-rank variant-7 answers bit for bit (a query's numerics do not depend on the shard it falls into on this variant either)
    with _ops.variant(7):
        r = nbv_step(*a2, occ_perms=P(g2), samples=T(g2["samples"]), group=dist.group.WORLD)
        c0, c1 = r["cam_range"]
        expect(torch.equal(r["occ"], s7_full["occ"]) and torch.equal(r["gains"], s7_full["gains"][c0:c1]), "V sharded step on variant 7")
        expect(torch.equal(r["max_gain"], s7_full["max_gain"]) and int(r["nbv_idx"]) == int(s7_full["nbv_idx"]) == int(g2["nbv_idx"]), "V decision")
        expect(not torch.equal(s7_full["occ"], s_full["occ"]), "V (variant 7 did run)")
        r = nbv_step_batch(*ab, occ_perms=permb, samples=ub, group=dist.group.WORLD)
        b0, b1 = r["cloud_range"]
        expect(torch.equal(r["occ"], s7_b3["occ"][b0:b1]) and torch.equal(r["gains"], s7_b3["gains"][b0:b1]), "V batch own clouds")
        expect(torch.equal(r["max_gain"], s7_b3["max_gain"]) and torch.equal(r["nbv_idx"], s7_b3["nbv_idx"]), "V batch decisions")
    # F: BASELINE config 4's scorer shape on two ranks: 100 000 points x 512 cameras, 256 cameras per rank, the decision through the
    # record exchange == the 1-rank arg-max over all 512
    from macarons_amd import ops
    from macarons_amd import dist as mdist
    gen = torch.Generator().manual_seed(404)
    pts4 = torch.cat([torch.rand(1, 100_000, 3, generator=gen) - 0.5, 0.1 + 0.9 * torch.rand(1, 100_000, 1, generator=gen)], -1).to(dev)
    harm4 = (torch.randn(1, 100_000, 64, generator=gen) * 0.5).to(dev)
    cams4 = torch.randn(1, 512, 3, generator=gen)
    cams4 = (1.5 * cams4 / cams4.norm(dim=-1, keepdim=True)).to(dev)
    full = ops.sh_coverage_gain(pts4, harm4, cams4)
    c0, c1 = mdist.shard_range(512, rank, 2)
    mine4 = ops.sh_coverage_gain(pts4, harm4, cams4[:, c0:c1].contiguous())
    v4, i4 = mdist.allgather_best(mine4, c0)
    ref4 = torch.max(full, dim=1)
    expect(torch.equal(mine4, full[:, c0:c1]) and torch.equal(v4, ref4.values) and torch.equal(i4, ref4.indices), "F config-4 scorer shards")
    # G: the MACARONS decision (config 5's regime) sharded over the two ranks -- field rows and neighbour cameras block-partitioned,
    # Cell.fill / SconeOcc draws from rank 0 -- on the reference's two-decision golden: bit-equal to the 1-rank decisions (which
    # tests/test_macarons_regime_gpu.py holds to the reference at 1e-4), state included
    torch.manual_seed(977 + rank)                                        # the ranks' own CPU generators disagree on purpose ...
    mac2, _ = macarons_decisions(dev, group=dist.group.WORLD)            # ... (macarons_decisions re-seeds: so shift rank 1's draws)
    for c in range(2):
        (r1, s1), (r2, s2) = mac1[c], mac2[c]
        k0, k1 = r2["cam_range"]
        expect((k0, k1) == ((0, 3) if rank == 0 else (3, 5)), f"G{c} cam_range")
        expect(torch.equal(r2["occ_probs"], r1["occ_probs"]) and torch.equal(r2["X_world"], r1["X_world"]), f"G{c} field")
        expect(torch.equal(r2["gains"], r1["gains"][k0:k1]), f"G{c} gains")
        expect(int(r2["next_idx"]) == int(r1["next_idx"]) == int(gmac[f"next_idx_{c}"]) and float(r2["max_gain"]) == float(r1["max_gain"]), f"G{c} decision")
        expect(all(torch.equal(s1[k], s2[k]) for k in s1), f"G{c} state")
    # H: group=None under an initialised process group is LOCAL (a data-parallel job whose ranks own different scenes must
    # not meet an implicit collective): the two ranks run DIFFERENT steps at the same time, each gets its own 1-rank answer
    if rank == 0:
        r = nbv_step(*a2, occ_perms=P(g2), samples=T(g2["samples"]))
        expect(r["cam_range"] == (0, 100) and torch.equal(r["gains"], s_full["gains"]) and int(r["nbv_idx"]) == int(s_full["nbv_idx"]), "H rank 0")
    else:
        r = nbv_step(*tiny, occ_perms=P(g1), samples=T(g1["samples"]))
        expect(torch.equal(r["occ"], s_tiny["occ"]) and torch.equal(r["max_gain"], s_tiny["max_gain"]), "H rank 1")
        macl, _ = macarons_decisions(dev)                                # a whole local MACARONS decision on rank 1 only
        expect(all(int(macl[c][0]["next_idx"]) == int(mac1[c][0]["next_idx"]) and torch.equal(macl[c][0]["gains"], mac1[c][0]["gains"])
                   for c in range(2)), "H local MACARONS decision")
    # E: hidden draws (nothing pinned): rank 0's reach rank 1 -> identical decisions on both ranks, single-cloud and batch
    torch.manual_seed(100 + rank)                                        # the ranks' own generators disagree on purpose
    r1 = nbv_step(*a2, group=dist.group.WORLD)
    r2 = nbv_step_batch(*ab, group=dist.group.WORLD)
    mine = torch.cat((r1["max_gain"].view(-1), r1["nbv_idx"].view(-1).float(), r2["max_gain"], r2["nbv_idx"].float(),
                      r1["occ"].sum().view(1)))
    mine = mine if backend == "nccl" else mine.cpu()
    both = [torch.empty_like(mine) for _ in range(2)]
    dist.all_gather(both, mine)
    expect(torch.equal(both[0], both[1]), "E ranks agree under hidden draws")

    t = torch.tensor([0 if fails else 1])
    if backend == "nccl":
        t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if fails:
        print(f"rank {rank} FAILED: {fails}", flush=True)
    if rank == 0 and int(t) == 1:
        print("TWO_RANK_OK", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(t) == 1 else 1)


if __name__ == "__main__":
    main()
