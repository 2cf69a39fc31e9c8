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
                                     (1, 50, 4097, 4), (2, 10, 2048, 1)])
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
