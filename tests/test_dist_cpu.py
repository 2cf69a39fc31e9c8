"""world_size-2 gloo tests (CPU) of the multi-GPU exchange logic: shard ranges, the all-gather of per-rank
(best gain, camera index) records and the all-gather of occupancy rows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from macarons_amd import dist as mdist
    try:
        # camera shards: every rank scores its block of a known gain vector
        gains_all = torch.tensor([[0.2, 0.9, 0.1, 0.9, 0.3, 0.05, 0.7], [0.5, 0.1, 0.1, 0.2, 0.8, 0.8, 0.0]])   # [B=2, C=7]
        c0, c1 = mdist.shard_range(7, rank, world)
        local = gains_all[:, c0:c1]
        best = torch.max(local, dim=1)
        v, i = mdist.allgather_argmax(best.values, best.indices + c0)
        ref = torch.max(gains_all, dim=1)                    # first occurrence wins ties, like the reference
        ok1 = torch.equal(v, ref.values) and torch.equal(i, ref.indices)
        v2, i2 = mdist.allgather_best(local.contiguous(), c0)      # the entry nbv_step uses (HIP kernels on a GPU, torch on CPU)
        ok1 = ok1 and torch.equal(v2, ref.values) and torch.equal(i2, ref.indices)
        # occupancy rows: uneven shards
        full = torch.arange(11 * 3, dtype=torch.float32).view(11, 3)
        q0, q1 = mdist.shard_range(11, rank, world)
        got = mdist.allgather_rows(full[q0:q1].clone(), 11)
        ok2 = torch.equal(got, full)
        # fewer items than ranks: rank 1 owns an empty shard and still joins both collectives (no hang, right answer)
        one = torch.tensor([[0.25]])
        e0, e1 = mdist.shard_range(1, rank, world)
        v3, i3 = mdist.allgather_best(one[:, e0:e1].contiguous(), e0)
        ok3 = float(v3[0]) == 0.25 and int(i3[0]) == 0
        rows = mdist.allgather_rows(torch.full((e1 - e0, 2), 7.0), 1)
        ok3 = ok3 and rows.shape == (1, 2) and bool((rows == 7.0).all())
        # a row of -inf / NaN keeps torch.max's answer (first index / first NaN)
        bad = torch.tensor([[float("-inf")] * 4, [1.0, float("nan"), 3.0, float("nan")]])
        b0, b1 = mdist.shard_range(4, rank, world)
        v4, i4 = mdist.allgather_best(bad[:, b0:b1].contiguous(), b0)
        rb = torch.max(bad, dim=1)
        ok3 = ok3 and int(i4[0]) == int(rb.indices[0]) and int(i4[1]) == int(rb.indices[1]) and bool(torch.isnan(v4[1]))
        # hidden draws: every rank draws its own randperms / uniforms, rank 0's reach everybody in one broadcast (odd and even counts)
        g = torch.Generator().manual_seed(100 + rank)
        perms = [torch.randperm(50, generator=g)[:20], torch.randperm(7, generator=g), torch.randperm(9, generator=g)[:3]]
        g0 = torch.Generator().manual_seed(100)
        want = [torch.randperm(50, generator=g0)[:20], torch.randperm(7, generator=g0), torch.randperm(9, generator=g0)[:3]]
        for n_u in (8, 5):
            u = torch.rand(n_u, 1, generator=torch.Generator().manual_seed(7 + rank))
            u0 = torch.rand(n_u, 1, generator=torch.Generator().manual_seed(7))
            got_p, got_u = mdist.broadcast_draws(perms, u)
            ok3 = ok3 and all(torch.equal(a, b) for a, b in zip(got_p, want)) and torch.equal(got_u, u0) and got_u.shape == (n_u, 1)
        only_p, none_u = mdist.broadcast_draws(perms, None)
        ok3 = ok3 and none_u is None and all(torch.equal(a, b) for a, b in zip(only_p, want))
        # the sharded MACARONS decision: SconeOcc's hidden draws of ALL (cell, chunk) jobs are rank 0's, in job order -- every rank ends
        # up with the permutations a 1-rank pass seeded like rank 0 would have drawn
        import contextlib, io
        from macarons_amd.networks import SconeOcc
        from macarons_amd.utility import macarons_utils as mu
        with contextlib.redirect_stdout(io.StringIO()):
            occ = SconeOcc()
        sizes = [700, 2300, 130, 5000]
        torch.manual_seed(500 + rank)                                        # the ranks' own CPU generators disagree
        got = mu._broadcast_job_perms(occ, sizes, torch.device("cpu"), None, rank)
        torch.manual_seed(500)
        want_j = [occ.draw_perms(m) for m in sizes]
        ok3 = ok3 and all(torch.equal(a, b) for ja, jb in zip(got, want_j) for a, b in zip(ja, jb))
        q.put((rank, bool(ok1), bool(ok2 and ok3), (c0, c1), (q0, q1)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partition():
    from macarons_amd.dist import shard_range
    for n in (1, 7, 8, 200, 512, 100_000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_allgather_argmax_and_rows_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(2)]
    for p in procs:
        p.join(30)
    assert all(r[1] and r[2] for r in res), res


def test_slice_index_arrays_matches_a_direct_build():
    """The index arrays a rank of the sharded MACARONS decision cuts out of rank 0's draws for ITS jobs (re-based to its own sub-list
    of clouds) are what building them for that sub-list alone gives."""
    import contextlib, io
    from macarons_amd.networks import SconeOcc
    from macarons_amd.utility import macarons_utils as mu
    with contextlib.redirect_stdout(io.StringIO()):
        occ = SconeOcc()
    sizes = [700, 2300, 130, 5000, 1024]
    Lg = occ.seq_len
    torch.manual_seed(3)
    perms = [occ.draw_perms(m) for m in sizes]

    def build(ps, ms):                                                       # the host construction of SconeOcc.forward_ragged
        off0 = np.concatenate(([0], np.cumsum(ms)))
        g_idx, g_len, i1, i2, o1, o2 = np.zeros((len(ms), Lg), np.int64), [], [], [], [0], [0]
        for j, (p0, p1, p2) in enumerate(ps):
            n0 = min(len(p0), Lg)
            g_idx[j, :n0] = off0[j] + p0[:n0].numpy(); g_idx[j, n0:] = off0[j]; g_len.append(n0)
            i1.append(off0[j] + p1.numpy()); i2.append(o1[-1] + p2.numpy())
            o1.append(o1[-1] + len(p1)); o2.append(o2[-1] + len(p2))
        T = lambda a, dt=torch.int64: torch.as_tensor(np.asarray(a), dtype=dt)
        return {"g_idx": T(g_idx.reshape(-1)), "g_len": T(g_len, torch.int32), "idx1": T(np.concatenate(i1)), "idx2": T(np.concatenate(i2)),
                "off1": T(o1), "off2": T(o2)}
    full = build(perms, sizes)
    for j0, j1 in ((0, 5), (1, 3), (2, 5), (4, 5)):
        cut = mu._slice_index_arrays(occ, full, sizes, j0, j1)
        want = build(perms[j0:j1], sizes[j0:j1])
        assert all(torch.equal(cut[k], want[k]) for k in want), (j0, j1)
