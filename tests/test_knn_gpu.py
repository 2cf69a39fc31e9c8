"""Parity of the HIP kNN kernel: bit-exact indices/distances vs the oracle convention, tie-aware vs the
reference goldens (the reference's own tie order is unspecified)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import knn

pytestmark = pytest.mark.gpu


def _run(dev, X, pc, k, sub=False):
    from macarons_amd import ops
    p, d, i = ops.knn_points(torch.from_numpy(X).to(dev), torch.from_numpy(pc).to(dev), k, sub)
    torch.cuda.synchronize()
    return p.cpu().numpy(), d.cpu().numpy(), i.cpu().numpy()


def test_golden(dev):
    g = golden("knn")
    for X, pc, idx, dist, atol in ((g["Xg"], g["pcg"], g["idx_g"], g["dist_g"], 1e-7),
                                   (g["Xr"], g["pcr"], g["idx_r"], g["dist_r"], 2e-5),
                                   (g["Xs"], g["pcs"], g["idx_s"], g["dist_s"], 2e-5)):
        p, d, i = _run(dev, X, pc, 16)
        assert i.dtype == np.int64
        po, do, io = knn.knn_points(X, pc, 16)
        assert np.array_equal(i, io)                 # bit-exact vs the oracle convention
        assert np.array_equal(d, do)
        assert np.array_equal(p, po)
        assert knn.tie_aware_index_match(i, d, idx, dist, X, pc, atol=atol)     # vs the reference


@pytest.mark.parametrize("B,Q,M,k", [(1, 1, 16, 16), (2, 300, 257, 16), (1, 1000, 5000, 16), (3, 129, 64, 8),
                                     (1, 50, 4097, 4), (2, 10, 2048, 1),
                                     (2, 300, 126, 16), (1, 1000, 256, 16), (3, 129, 17, 16), (1, 70000, 160, 16)])   # small clouds (SconeOcc's coarsest scale)
def test_ragged_vs_oracle(dev, B, Q, M, k):
    rng = np.random.default_rng(Q * 7 + M)
    X = rng.uniform(-.5, .5, (B, Q, 3)).astype(np.float32)
    pc = rng.uniform(-.5, .5, (B, M, 3)).astype(np.float32)
    pc[:, 3] = pc[:, 1]                                     # exact duplicate point -> exact tie
    p, d, i = _run(dev, X, pc, k, sub=True)
    po, do, io = knn.knn_offsets(X, pc, k)
    assert np.array_equal(i, io) and np.array_equal(d, do) and np.array_equal(p, po)
    assert np.all(np.diff(d, axis=-1) >= 0)                 # ascending


def test_k_larger_than_m_raises(dev):
    from macarons_amd import ops, _lib
    with pytest.raises(_lib.MacaronsHipError):
        ops.knn_points(torch.zeros(1, 4, 3, device=dev), torch.zeros(1, 8, 3, device=dev), 16)


def test_large_properties(dev):
    """Q=100k, M=10k (BASELINE step size): a sample of queries against the oracle + invariants."""
    rng = np.random.default_rng(5)
    X = rng.uniform(-.5, .5, (1, 100_000, 3)).astype(np.float32)
    pc = rng.uniform(-.5, .5, (1, 10_240, 3)).astype(np.float32)
    p, d, i = _run(dev, X, pc, 16, sub=True)
    assert np.all(np.diff(d, axis=-1) >= 0) and i.min() >= 0 and i.max() < 10_240
    sel = rng.choice(100_000, 512, replace=False)
    po, do, io = knn.knn_offsets(X[:, sel], pc, 16)
    assert np.array_equal(i[:, sel], io) and np.array_equal(d[:, sel], do) and np.array_equal(p[:, sel], po)
    # offsets consistent with indices
    assert np.array_equal(p, pc[0][i[0]][None] - X[:, :, None, :])


def _grid_cases():
    """Clouds and query sets that stress the grid-pruned search (K1-grid, knn.hip): the output must stay that of the brute-force
    convention whatever the geometry does to the pruning."""
    rng = np.random.default_rng(11)
    u = lambda *s: rng.uniform(-.5, .5, s).astype(np.float32)
    cases = {}
    cases["uniform"] = (u(1, 3000, 3), u(1, 4096, 3))
    # heavy ties: both sets on a coarse lattice (dozens of candidates at exactly the 16th distance)
    lat = lambda n: (rng.integers(0, 9, (1, n, 3)) / 8.0 - 0.5).astype(np.float32)
    cases["lattice_ties"] = (lat(1500), lat(3000))
    # a surface-like cloud (thin spherical shell) with queries filling the box: most queries are far from every candidate
    d = rng.normal(size=(1, 5000, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cases["shell_far_queries"] = (u(1, 2500, 3) * 3.0, (0.3 * d).astype(np.float32))
    # two tight clusters far apart + queries between them (the seed sub-tiles are far from the true neighbours of some queries)
    c = np.concatenate([u(1, 1200, 3) * 0.01 + 5.0, u(1, 1200, 3) * 0.01 - 5.0], axis=1)
    cases["two_clusters"] = (np.concatenate([u(1, 700, 3) * 12.0, c[:, ::7] + 1e-3], axis=1).astype(np.float32), c.astype(np.float32))
    # degenerate boxes: every candidate identical (all cells collapse), then a planar cloud (zero extent on one axis)
    cases["all_identical"] = (u(1, 300, 3), np.full((1, 1100, 3), 0.25, np.float32))
    flat = u(1, 2048, 3); flat[..., 2] = 0.125
    cases["planar"] = (u(1, 1000, 3), flat)
    # not a multiple of 32 candidates / of 128 queries, a batch of clouds, a cloud far from the origin (filter guard band)
    cases["ragged_batch"] = (u(3, 333, 3), u(3, 1031, 3))
    cases["offset_origin"] = (u(1, 777, 3) + 100.0, u(1, 2000, 3) + 100.0)
    cases["one_query"] = (u(1, 1, 3), u(1, 1024, 3))
    # every query near the centre of a sphere: all candidates are (nearly) equidistant, every group of 32 queries is "heavy" and is
    # parked for the split pass -- 500 groups for the 384 parking slots, so the groups that find no slot finish on their own as well
    sph = rng.normal(size=(1, 4096, 3)); sph /= np.linalg.norm(sph, axis=-1, keepdims=True)
    cases["sphere_centre"] = (u(1, 16000, 3) * 0.02, (0.3 * sph).astype(np.float32))
    cases["sphere_centre_batch"] = (u(2, 640, 3) * 0.05, np.stack([(0.3 * sph[0]).astype(np.float32), (0.2 * sph[0, ::-1]).astype(np.float32)]))
    return cases


@pytest.mark.parametrize("name", list(_grid_cases()))
def test_grid_pruned_search_is_exact(dev, name):
    X, pc = _grid_cases()[name]
    p, d, i = _run(dev, X, pc, 16, sub=True)
    po, do, io = knn.knn_offsets(X, pc, 16)
    assert np.array_equal(i, io), name
    assert np.array_equal(d, do) and np.array_equal(p, po), name


def test_grid_pruned_equals_brute_force_kernels(dev, monkeypatch):
    """Same call through mcr_knn_points (brute force, MFMA filter) and mcr_knn_points_grid: bit-identical outputs at the step's size."""
    import ctypes
    from macarons_amd import ops
    from macarons_amd._lib import lib, check, c_i64, c_int
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.uniform(-.5, .5, (2, 20_000, 3)).astype(np.float32)).to(dev)
    pc = torch.from_numpy(rng.normal(0, .2, (2, 10_240, 3)).astype(np.float32)).to(dev)
    p1, d1, i1 = ops.knn_points(X, pc, 16, True)
    i0 = torch.empty_like(i1); d0 = torch.empty_like(d1); p0 = torch.empty_like(p1)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    check(lib().mcr_knn_points(ptr(X), ptr(pc), ptr(i0), ptr(d0), ptr(p0), c_i64(2), c_i64(20_000), c_i64(10_240), c_int(16), c_int(1),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mcr_knn_points")
    torch.cuda.synchronize()
    assert torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(p0, p1)


@pytest.mark.parametrize("M", [1023, 1024, 1025, 16384, 16385])
def test_grid_routing_boundaries(dev, M):
    """Either side of the sizes at which mcr_knn_points_grid switches between the brute-force kernels and the grid search
    (1024 <= M <= 16384): the same exact answer, on a clustered cloud with far queries (so that groups get parked where the grid runs)."""
    rng = np.random.default_rng(M)
    c = rng.normal(size=(1, M, 3)); c /= np.linalg.norm(c, axis=-1, keepdims=True)
    pc = (0.3 * c + 0.001 * rng.normal(size=(1, M, 3))).astype(np.float32)
    X = rng.uniform(-.5, .5, (1, 1500, 3)).astype(np.float32)
    X[0, :200] *= 0.05                                              # queries near the centre of the shell: heavy groups
    p, d, i = _run(dev, X, pc, 16, sub=True)
    po, do, io = knn.knn_offsets(X, pc, 16)
    assert np.array_equal(i, io) and np.array_equal(d, do) and np.array_equal(p, po)


def test_segmented_search_equals_the_per_job_search(dev):
    """mcr_knn_offsets_segmented (J jobs of different sizes in one launch: the ragged occupancy pass) == mcr_knn_points job by job, bit
    for bit -- with and without the scratch that lets a launch with few query blocks split every job's candidates over several
    workgroups and merge their (d2, index) lists (the merge must keep the tie order: clouds on a 2^-3 grid with duplicated points,
    i.e. hundreds of equal distances per query), incl. a job of one query, a cloud of exactly 16 points and one whose candidate
    slices come out ragged."""
    from macarons_amd import ops
    rng = np.random.default_rng(31)
    sizes_m, sizes_q = [16, 5000, 1025, 12001, 33, 4097], [1, 300, 129, 40, 128, 7]
    clouds = [(rng.integers(-4, 5, (m, 3)) / 8.0).astype(np.float32) for m in sizes_m]
    clouds[3] = rng.uniform(-.5, .5, (sizes_m[3], 3)).astype(np.float32)             # one real-valued job
    xs = [(rng.integers(-8, 9, (q, 3)) / 16.0).astype(np.float32) for q in sizes_q]
    X, pc = torch.from_numpy(np.concatenate(xs)).to(dev), torch.from_numpy(np.concatenate(clouds)).to(dev)
    split = ops.knn_offsets_segmented(X, pc, sizes_m, sizes_q, split=True)
    plain = ops.knn_offsets_segmented(X, pc, sizes_m, sizes_q, split=False)
    assert torch.equal(split, plain)
    r0 = 0
    for c, x in zip(clouds, xs):
        pts, _, _ = ops.knn_points(torch.from_numpy(x[None]).to(dev), torch.from_numpy(c[None]).to(dev), 16, subtract_query=True)
        assert torch.equal(pts[0], split[r0:r0 + len(x)]), len(c)
        po, _, _ = knn.knn_offsets(x[None], c[None], 16)                       # and the oracle convention itself
        assert np.array_equal(po[0], split[r0:r0 + len(x)].cpu().numpy()), len(c)
        r0 += len(x)
