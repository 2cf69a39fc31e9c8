"""The batched hidden draws (csrc_torch/macarons_torch.cpp: randperm_prefixes, scone_occ_draws) make the reference's torch.randperm
calls -- same values, same generator state afterwards -- and lay the indices out as SconeOcc.forward_ragged / Scene.fill_cells need."""
import contextlib
import importlib
import io

import numpy as np
import torch


def test_randperm_prefixes_is_the_loop_of_torch_randperm():
    import macarons_amd.torch_ops  # noqa: F401
    for seed, spec in ((9, ((10, 4), (5000, 100000), (7, 7), (1, 1), (20000, 1000))), (1, ((3, 0), (2, 5)))):
        torch.manual_seed(seed)
        want = torch.cat([torch.randperm(n)[:k] for n, k in spec])
        after = torch.randperm(11)
        torch.manual_seed(seed)
        got = torch.ops.macarons.randperm_prefixes([n for n, _ in spec], [k for _, k in spec])
        assert torch.equal(got, want) and torch.equal(torch.randperm(11), after)


def test_scone_occ_draws_are_draw_perms_job_by_job():
    import macarons_amd.torch_ops  # noqa: F401
    M = importlib.import_module("macarons_amd.networks.SconeOcc")
    with contextlib.redirect_stdout(io.StringIO()):
        occ = M.SconeOcc()
    sizes = [700, 2300, 130, 5000, 1024, 27000, 65, 2048, 2049]
    sz = [occ.scale_sizes(m) for m in sizes]
    J, Lg = len(sizes), occ.seq_len
    torch.manual_seed(3)
    perms = [occ.draw_perms(m) for m in sizes]                     # SconeOcc.py:269, :311 -- three torch.randperm per forward
    after = torch.randperm(5)
    off0 = np.concatenate(([0], np.cumsum(sizes)))
    g_idx, g_len, idx1, idx2, off1, off2 = np.zeros((J, Lg), np.int64), np.zeros(J, np.int64), [], [], [0], [0]
    for j, (p0, p1, p2) in enumerate(perms):
        p0, p1, p2 = np.asarray(p0), np.asarray(p1), np.asarray(p2)
        assert len(p1) == sz[j][1] and len(p2) == sz[j][2]
        n0 = min(len(p0), Lg)
        g_idx[j, :n0] = off0[j] + p0[:n0]; g_idx[j, n0:] = off0[j]; g_len[j] = n0
        idx1.append(off0[j] + p1); idx2.append(off1[-1] + p2)
        off1.append(off1[-1] + len(p1)); off2.append(off2[-1] + len(p2))
    want = np.concatenate([g_idx.reshape(-1), np.concatenate(idx1), np.concatenate(idx2), off1, off2, g_len])
    torch.manual_seed(3)
    got = torch.ops.macarons.scone_occ_draws([s[0] for s in sz], [s[1] for s in sz], [s[2] for s in sz], Lg)
    assert np.array_equal(got.numpy(), want) and torch.equal(torch.randperm(5), after)


def test_native_bin_permutation_equals_the_python_restatement():
    """torch.ops.macarons.view_space_bins (C++: the same ATen operators in the same order, one dispatcher call) returns the bins of
    scone_utils.view_space_bin_indices for random and for axis-aligned rotations (directions on bin boundaries)."""
    import torch
    from macarons_amd.utility import scone_utils as su
    if not su._native_bins():
        import pytest
        pytest.skip("C++ extension not built")
    torch.manual_seed(5)
    for n_elev, n_azim in ((7, 14), (5, 10)):
        for i in range(300):
            R = torch.linalg.qr(torch.randn(3, 3))[0].float()
            if i % 5 == 0:
                R = torch.eye(3)[torch.randperm(3)] * torch.tensor([1., -1., 1.])[torch.randperm(3)]
            a = su.view_space_bin_permutation(R, n_elev, n_azim, "cpu")
            x_ref = su._REF_DIRECTIONS[(n_elev, n_azim)]
            b = su.view_space_bin_indices((x_ref @ R.view(3, 3).T).reshape(-1, 3), n_elev, n_azim)
            assert torch.equal(a, b), (n_elev, n_azim, i)


def test_native_field_jobs_equal_the_numpy_tables():
    """torch.ops.macarons.field_jobs (C++ loop) builds the (cell, chunk) job tables of the occupancy-field pass element for element
    as the numpy restatement in macarons_utils.compute_scene_occupancy_probability_field does: random grids, empty cells, cells that
    do not run (too few surface points, no queries, never visited), several chunks per cell."""
    import numpy as np
    import torch
    from macarons_amd.utility import macarons_utils as mu
    if not mu._native_field_jobs():
        import pytest
        pytest.skip("C++ extension not built")
    rng = np.random.default_rng(3)
    for trial in range(40):
        n_cells = int(rng.integers(1, 80))
        chunk = int(rng.choice([7, 50, 20000]))
        k = int(rng.choice([1, 16]))
        visit = (rng.random(n_cells) < 0.7).astype(np.int64)
        counts = (rng.integers(0, 300, n_cells) * (rng.random(n_cells) < 0.8)).astype(np.int64)
        sel_off = np.concatenate(([0], np.cumsum(counts), [counts.sum()])).astype(np.int64)        # n_cells + 2 entries
        n_oof = int(rng.integers(0, 50))
        hostc = np.concatenate((visit, [0], counts, [0], sel_off, [n_oof])).astype(np.int64)
        s_len = (rng.integers(0, 400, n_cells) * (rng.random(n_cells) < 0.7)).astype(np.int64)
        s_off = np.concatenate(([0], np.cumsum(s_len))).astype(np.int64)
        nbm = np.full((n_cells, 27), -1, np.int64)
        for c in range(n_cells):
            nb = np.unique(rng.integers(0, n_cells, int(rng.integers(1, 28))))
            nbm[c, :nb.size] = nb
        xf_all = rng.standard_normal((n_cells, 20)).astype(np.float32)
        perm = rng.permutation(98).astype(np.int32)
        raw_t, meta_t = torch.ops.macarons.field_jobs(torch.from_numpy(hostc), torch.from_numpy(s_off), torch.from_numpy(nbm),
                                                      torch.from_numpy(xf_all), torch.from_numpy(perm), n_cells, chunk, k)
        meta = meta_t.numpy()
        # ---- the numpy restatement (macarons_utils.py)
        s_start = s_off[:-1]
        nb_len = np.where(nbm >= 0, s_len[np.maximum(nbm, 0)], 0)
        m_cell = nb_len.sum(1)
        run = (visit != 0) & (m_cell > 2 * 2 * k) & (counts > 0)
        cells_run = np.nonzero(run)[0]
        n_chunks = -(-counts[cells_run] // chunk)
        job_cell = np.repeat(cells_run, n_chunks)
        J = int(job_cell.size)
        lo = (np.arange(J) - np.repeat(np.cumsum(n_chunks) - n_chunks, n_chunks)) * chunk
        job_q = np.minimum(chunk, counts[job_cell] - lo).astype(np.int64)
        job_m = m_cell[job_cell].astype(np.int64)
        q_start = np.concatenate(([0], np.cumsum(job_q))).astype(np.int64)
        m_start = np.concatenate(([0], np.cumsum(job_m))).astype(np.int64)
        jt = np.stack((sel_off[job_cell] + lo, q_start[:-1], m_start[:-1], np.zeros(J, np.int64)), 1).astype(np.int64) if J else np.zeros((0, 4), np.int64)
        seg_len = nb_len[job_cell]
        keep = seg_len > 0
        seg_l = seg_len[keep]
        st_ = np.stack((s_start[nbm[job_cell][keep]], np.cumsum(seg_l) - seg_l, np.nonzero(keep)[0], np.zeros(seg_l.size, np.int64)), 1).astype(np.int64)
        raw = np.concatenate((jt.reshape(-1).view(np.uint8), st_.reshape(-1).view(np.uint8),
                              np.ascontiguousarray(xf_all[job_cell]).reshape(-1).view(np.uint8), perm.view(np.uint8)))
        assert int(meta[0]) == J and int(meta[1]) == st_.shape[0] and int(meta[2]) == int(q_start[-1]) and int(meta[3]) == int(m_start[-1]) and int(meta[4]) == n_oof
        assert np.array_equal(meta[5:5 + J], job_q) and np.array_equal(meta[5 + J:5 + 2 * J], job_m)
        assert np.array_equal(meta[5 + 2 * J:6 + 3 * J], q_start) and np.array_equal(meta[6 + 3 * J:7 + 4 * J], m_start)
        assert np.array_equal(raw_t.numpy(), raw), trial


def test_native_ragged_tables_equal_the_numpy_tables():
    """torch.ops.macarons.ragged_tables == the numpy tables of SconeOcc.forward_ragged_begin (cloud offsets, query blocks, row -> job)."""
    import numpy as np
    import torch
    import importlib
    so = importlib.import_module("macarons_amd.networks.SconeOcc")     # (the package re-exports the CLASS under the module's name)
    if not so._native_ragged_tables():
        import pytest
        pytest.skip("C++ extension not built")
    rng = np.random.default_rng(0)
    for _ in range(100):
        J = int(rng.integers(1, 30)); cs = rng.integers(16, 5000, J).tolist(); qs = rng.integers(0, 900, J).tolist(); rows = int(rng.choice([64, 128]))
        for wr in (True, False):
            a = torch.ops.macarons.ragged_tables(cs, qs, rows, wr).numpy()
            off0 = np.concatenate(([0], np.cumsum(cs))).astype(np.int64); q = np.asarray(qs, np.int64); q0 = np.concatenate(([0], np.cumsum(q)))
            nb = -(-q // rows); bj = np.repeat(np.arange(J, dtype=np.int64), nb)
            b_in = np.arange(int(nb.sum()), dtype=np.int64) - np.repeat(np.cumsum(nb) - nb, nb)
            blocks = np.stack((bj, q0[bj] + b_in * rows, np.minimum(rows, q[bj] - b_in * rows), np.zeros_like(bj)), 1)
            parts = [off0, blocks.reshape(-1)] + ([np.repeat(np.arange(J, dtype=np.int64), q)] if wr else [])
            assert np.array_equal(a, np.concatenate(parts))


def test_native_field_prepare_equals_the_python_restatement():
    """torch.ops.macarons.field_prepare (the cells' prediction-box transforms + the view-space bin permutation in one C++ call over the
    same ATen operators) == macarons_utils._field_prepare's tensor code, bit for bit."""
    import torch
    from macarons_amd.utility import scone_utils as su, macarons_utils as mu
    if not (mu._native_field_jobs() and hasattr(torch.ops.macarons, "field_prepare")):
        import pytest
        pytest.skip("C++ extension not built")
    torch.manual_seed(1)
    su.view_space_bin_permutation(torch.eye(3), 7, 14, "cpu")
    x_ref = su._REF_DIRECTIONS[(7, 14)]
    for i in range(200):
        R = torch.linalg.qr(torch.randn(3, 3))[0].float()
        if i % 5 == 0:
            R = torch.eye(3)[torch.randperm(3)] * torch.tensor([1., -1., 1.])[torch.randperm(3)]
        Mv = torch.eye(4); Mv[:3, :3] = R; Mv[3, :3] = torch.randn(3) * 20
        n = int(torch.randint(1, 100, (1,)))
        cw, dg, pns = torch.randn(n, 3) * 10, torch.rand(n) * 5 + 1, 1.5
        a, b = torch.ops.macarons.field_prepare(Mv, cw, dg, x_ref, pns, 7, 14)
        cen_h = (torch.cat((cw, torch.ones(n, 1)), 1) @ Mv)[:, :3]
        inv_h = (1.0 / (pns * dg)).float()
        xf = torch.cat((Mv.reshape(1, 16).expand(n, -1), cen_h, inv_h.view(n, 1)), 1).contiguous()
        perm = su.view_space_bin_indices((x_ref @ Mv[:3, :3].contiguous().view(3, 3).T).reshape(-1, 3), 7, 14).to(torch.int32)
        assert torch.equal(a, xf) and torch.equal(b, perm), i
        if i % 10 == 0:                 # the worker-thread form: same tensors; the caller may rewrite its matrix once the ticket is out
            Mv2 = Mv.clone()
            t = torch.ops.macarons.field_prepare_async(Mv2, cw, dg, x_ref, pns, 7, 14)
            Mv2.zero_()
            a2, b2 = torch.ops.macarons.field_prepare_wait(t)
            assert torch.equal(a2, a) and torch.equal(b2, b), i
    import pytest
    tickets = [torch.ops.macarons.field_prepare_async(Mv, cw, dg, x_ref, pns, 7, 14) for _ in range(4)]     # collected out of order
    for t in reversed(tickets):
        a2, b2 = torch.ops.macarons.field_prepare_wait(t)
        assert torch.equal(a2, a) and torch.equal(b2, b)
    with pytest.raises(RuntimeError):
        torch.ops.macarons.field_prepare_wait(tickets[0])          # a ticket is good for one collection
    t = torch.ops.macarons.field_prepare_async(torch.zeros(3, 3), cw, dg, x_ref, pns, 7, 14)
    with pytest.raises(RuntimeError):                               # the job's error surfaces at the wait
        torch.ops.macarons.field_prepare_wait(t)
