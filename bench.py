#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native SCONE coverage-gain hot path.

Metric (BASELINE.json): candidate-camera coverage-gain evals/sec (100k pts, 200 cams); NBV step p50 latency.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Two quantities are reported:
  (A) `value`: scorer throughput.  A "step" = one pass of the scorer (SconeVis.compute_coverage_gain semantics) over one synthetic cloud of
100 000 points x 200 candidate cameras per GPU, inputs already resident in HBM, followed (N>1) by the
all-gather of each rank's (best gain, camera index).  One (cloud, camera) pair scored = one eval.
Weak scaling: each rank owns a disjoint shard of 200 candidate cameras of the same cloud (the reference
scores all cameras on one GPU).

  (B) `nbv_step`: p50 latency of the full SCONE NBV decision (macarons_amd.nbv.nbv_step: view state + harmonics on
      Q = 100k proxy points -> SconeOcc vs M = 10 240 surface points -> sample 2048 -> SconeVis -> gains over C = 200
      cameras -> arg-max), device-synchronised per iteration, random-init weights; with N GPUs the queries and the
      cameras are sharded (strong scaling of one decision).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — dominant kernel (sh_gain_kernel) algorithmic flop rate vs the fp32 vector peak
  cpu_baseline — the plain-C port of the reference scorer (oracle/csrc) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 370.0          # algorithmic flop per (point, camera) pair
BYTES_PER_POINT = 268.0        # 12 B xyz + 256 B coefficients, read once per cloud
PEAK_FP32_TFLOPS = 157.3       # MI355X fp32 vector (= fp32 MFMA) peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def make_inputs(N, C, seed, device, cam_offset=0, n_cam_total=None):
    """Synthetic workload of SURVEY §8d: uniform points in [-0.5,0.5]^3 + occupancy U(0.1,1), coefficients
    N(0,0.5^2), cameras on the radius-1.5 sphere."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    pts = torch.cat([torch.rand(1, N, 3, generator=g) - 0.5, 0.1 + 0.9 * torch.rand(1, N, 1, generator=g)], -1)
    harm = torch.randn(1, N, 64, generator=g) * 0.5
    n_tot = n_cam_total or C
    cams = torch.randn(1, n_tot, 3, generator=g)
    cams = 1.5 * cams / cams.norm(dim=-1, keepdim=True)
    cams = cams[:, cam_offset:cam_offset + C].contiguous()
    return pts.to(device), harm.to(device), cams.to(device)


def cpu_baseline(pts, harm, cams):
    """Time the C port of the reference scorer on the host cores for a bounded sample of the workload
    (the same cloud, the first cams.shape[1] cameras)."""
    from oracle import cport
    p, h, c = pts.cpu().numpy(), harm.cpu().numpy(), cams.cpu().numpy()
    N, C_sample = p.shape[1], c.shape[1]
    cport.coverage_gain(p[:, :256], h[:, :256], c)        # warm-up / build
    reps, t0 = 0, time.perf_counter()
    while True:                                           # ~10 s of CPU work, at least one full pass
        g, nthreads = cport.coverage_gain(p, h, c)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 50:
            break
    return {"value": reps * C_sample / dt, "unit": "evals/s", "cores": int(nthreads), "kind": "port",
            "sample": f"C port (oracle/csrc/scorer_port.c, OpenMP) of the reference scorer on the same cloud: "
                      f"N={N} points x {C_sample} cameras x {reps} passes, {dt:.2f} s wall; "
                      f"host has {os.cpu_count()} cores"}, g


def pmc_traffic_bytes():
    """HBM read bytes per launch of the scorer kernel from the committed PMC pass (None if the profile is absent)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_scorer_pmc.json")) as f:
            return float(json.load(f)["hbm_read_bytes_per_launch_corrected"])
    except Exception:
        return None


def measure_nbv_step(dev, rank, world, args):
    """(B) p50 latency of one NBV decision at Q=100k / M=10240 / C=200 (sharded over `world` GPUs)."""
    from macarons_amd.networks import SconeVis, SconeOcc
    from macarons_amd.nbv import nbv_step, ViewStateGrid
    import io, contextlib
    torch.manual_seed(7)                                   # identical random-init weights on every rank
    with contextlib.redirect_stdout(io.StringIO()):
        occ, vis = SconeOcc(), SconeVis()
    with torch.no_grad():
        occ.linear3.bias += 0.5                            # untrained occupancies must pass min_occ
    occ, vis = occ.to(dev).eval(), vis.to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(4321)
    Q, M, C = 100_000, 10_240, args.cams
    d = torch.randn(M, 3, generator=g)
    pc = (d / d.norm(dim=1, keepdim=True) * torch.tensor([0.35, 0.25, 0.3]) + 0.002 * torch.randn(M, 3, generator=g))[None].to(dev)
    X = (torch.rand(1, Q, 3, generator=g) - 0.5).to(dev)
    cams = torch.randn(C, 3, generator=g)
    cams = (1.5 * cams / cams.norm(dim=1, keepdim=True)).to(dev)
    X_view = cams[:3].contiguous()
    u = torch.rand(2048, generator=g).to(dev)
    grid = ViewStateGrid(dev)
    torch.manual_seed(11)
    perms = [p.to(dev) for p in occ.draw_perms(M)]       # the three randperm draws of SconeOcc.forward, pinned and resident
    group = torch.distributed.group.WORLD if torch.distributed.is_initialized() else None     # shards Q and C over the ranks
    times = []
    for it in range(10 + args.nbv_iters):
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        r = nbv_step(occ, vis, pc, X, X_view, cams, grid, occ_perms=perms, samples=u, group=group)
        int(r["nbv_idx"])                                  # the decision reaches the host
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tw = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tw, op=torch.distributed.ReduceOp.MAX)
            dt = float(tw.item())
        if it >= 10:
            times.append(dt)
    p50 = float(np.median(times))
    return {"p50_ms": p50 * 1e3, "p90_ms": float(np.percentile(times, 90)) * 1e3, "evals_per_s": C / p50, "iters": len(times),
            "config": {"proxy_points": Q, "surface_points": M, "cams": C, "seq_len": 2048, "dtype": "f32",
                       "parallelism": f"query+camera shard x{world}"},
            "algorithmic_TFLOP": 26.5e6 * Q / 1e12 + 0.0037 + 0.0137, "nbv_idx": int(r["nbv_idx"]), "n_unique": int(r["n_unique"])}


def measure_local_pct(dev):
    """Roofline of the dominant kernel of the NBV step (fused local transformer): HIP events around back-to-back
    launches on the launch stream.  Default kernel = split-precision bf16x6 (local_pct5.hip): every algorithmic
    fp32 multiply-add runs as 6 bf16 MFMA multiply-adds, so the matrix pipe executes 6x the algorithmic GEMM flops and
    is priced against the dense bf16 MFMA peak; the exact-fp32-MFMA kernel (local_pct.hip) is timed beside it."""
    import ctypes
    from macarons_amd import ops, _lib
    from macarons_amd.networks import SconeOcc
    from macarons_amd.networks.packing import pack_local_pct
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        occ = SconeOcc().to(dev)
    S = 16384
    offs = torch.randn(S, 16, 3, device=dev) * 0.05
    gemm_flops = S * 16 * 0.49e6                            # 0.49 MFLOP per token in linear layers
    alg_flops = gemm_flops + S * 0.164e6                    # + 16x16 attention per query
    L = _lib.lib()
    default_variant = L.mcr_get_local_pct_variant()
    out = {}
    for v in sorted({1, default_variant}):
        L.mcr_set_local_pct_variant(ctypes.c_int(v))
        blob = pack_local_pct(occ.local_transformers[0], v)
        for _ in range(3):
            ops.local_pct_forward(offs, blob)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.local_pct_forward(offs, blob)
        e1.record()
        torch.cuda.synchronize()
        out[v] = e0.elapsed_time(e1) / n
    L.mcr_set_local_pct_variant(ctypes.c_int(default_variant))
    ms = out[default_variant]
    if default_variant in (3, 4, 5):
        executed = 6.0 * gemm_flops / (ms * 1e-3) / 1e12
        peak, kern, note = 2500.0, f"local_pct{default_variant}_kernel", "bf16 MFMA dense peak 2.5 PFLOP/s; 6 bf16 MFMAs per exact fp32 product"
    else:
        executed = gemm_flops / (ms * 1e-3) / 1e12
        peak, kern, note = PEAK_FP32_TFLOPS, "local_pct_kernel", "fp32 MFMA (v_mfma_f32_32x32x2_f32) peak = 157.3 TFLOP/s"
    return {"kernel": kern, "bound": "mfma", "achieved": executed, "peak": peak, "unit": "TFLOP/s", "frac": executed / peak,
            "traffic": None, "device_ms_per_launch": ms, "queries_per_launch": S,
            "algorithmic_fp32_TFLOPs": alg_flops / (ms * 1e-3) / 1e12, "note": note,
            "exact_fp32_mfma_variant": {"kernel": "local_pct_kernel", "device_ms_per_launch": out[1],
                                        "achieved": gemm_flops / (out[1] * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS,
                                        "frac": gemm_flops / (out[1] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--cams", type=int, default=200)
    ap.add_argument("--waves-per-simd", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nbv", action="store_true", help="skip the NBV-step latency measurement")
    ap.add_argument("--nbv-iters", type=int, default=50)
    ap.add_argument("--nbv-multi", action="store_true", help="also time the query/camera-sharded NBV step when --gpus > 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("MCR_BENCH_FORCE_DIST"))      # the env knob exercises the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from macarons_amd import ops
    N, C = args.points, args.cams
    # every rank: same cloud, its own shard of C cameras out of world*C (weak scaling)
    pts, harm, cams = make_inputs(N, C, 1234, dev, cam_offset=rank * C, n_cam_total=world * C)

    pipe = None
    if use_dist:
        # the arg-max exchanges are batched (16 decisions per all-gather) and run on a side stream under the next scoring passes
        from macarons_amd import dist as mdist
        pipe = mdist.PipelinedBest(1, dev, batch=16, depth=3)

    def step():
        gains = ops.sh_coverage_gain(pts, harm, cams, True, args.waves_per_simd)
        if pipe is not None:
            return pipe.submit(gains, rank * C)
        return ops.best_record(gains)                        # [B,2] = (max gain, arg-max camera): the decision (torch.max semantics)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    if pipe is not None:
        pipe.flush()
        out = pipe.result(out)                               # the last decision (and with it all earlier ones) is complete
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    dev_ms = ev0.elapsed_time(ev1)                 # HIP events on the launch stream (torch current stream)

    # the sharded NBV step at N > 1 is opt-in: the contract line is the scorer step above
    nbv = measure_nbv_step(dev, rank, world, args) if (not args.no_nbv and (world == 1 or args.nbv_multi)) else None
    lp = measure_local_pct(dev) if (rank == 0 and not args.no_nbv) else None

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        evals_per_s = world * C * args.steps / wall
        kern_ms = dev_ms / args.steps              # per-launch device time of the scorer pass
        achieved = N * C * FLOP_PER_PAIR / (kern_ms * 1e-3) / 1e12
        res = {
            "metric": "candidate-camera coverage-gain evals/sec (100k pts, 200 cams)",
            "value": evals_per_s, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"scorer: B=1 cloud x N={N} points x C={C} cameras per GPU "
                                   f"(BASELINE headline 100k pts / 200 cams), inputs resident in HBM",
                       "points": N, "cams_per_gpu": C, "parallelism": f"camera-shard x{world}"},
            "roofline": {"bound": "valu-fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": pmc_traffic_bytes(),
                         "traffic_source": "profiles/r01_scorer_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 per MI355X_MICROARCH.md, own pass)",
                         "algorithmic_bytes": N * BYTES_PER_POINT,
                         "kernel": "sh_gain_kernel<true>", "device_ms_per_launch": kern_ms,
                         "hbm_algorithmic_GBs": N * BYTES_PER_POINT / (kern_ms * 1e-3) / 1e9},
        }
        if nbv is not None:
            res["nbv_step"] = nbv
        if lp is not None:
            res["roofline_nbv_dominant"] = lp
        if not args.no_cpu_baseline and world == 1:
            n_s = min(C, max(24, os.cpu_count() or 1))
            cb, g_cpu = cpu_baseline(pts, harm, cams[:, :n_s].contiguous())
            res["cpu_baseline"] = cb
            g_gpu = ops.sh_coverage_gain(pts, harm, cams[:, :n_s].contiguous()).cpu().numpy()
            res["cpu_baseline"]["max_rel_diff_vs_gpu"] = float(np.abs(g_gpu - g_cpu).max() / np.abs(g_cpu).max())
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
