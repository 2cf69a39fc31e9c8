"""k nearest surface points of every query (numpy restatement).  TEST INFRASTRUCTURE ONLY.

Restates macarons/utility/utils.py:1497-1509 get_knn_points:
    dists = torch.cdist(X, pc); topk(k, largest=False) (ascending); knn_gather(pc, idx)
and the offset step of macarons/networks/SconeOcc.py:297-298 (local_pc - x).

torch.cdist's large-matrix path uses |x|^2+|y|^2-2x.y (values differ from the direct form by ~5e-6) and
topk's tie order is unspecified, so "bit-exact indices" is only well defined away from (near-)ties
(SURVEY §7).  The oracle fixes the convention the HIP kernel implements exactly:
    d2 = (dx*dx + dy*dy) + dz*dz   in fp32, each product and sum rounded (no FMA contraction),
    ascending by (d2, index)  — ties go to the lower index,  dists = sqrt(d2) correctly rounded.
"""
import numpy as np


def d2_matrix(X, pc):
    X = np.asarray(X, np.float32)
    pc = np.asarray(pc, np.float32)
    dx = X[:, None, 0] - pc[None, :, 0]
    dy = X[:, None, 1] - pc[None, :, 1]
    dz = X[:, None, 2] - pc[None, :, 2]
    return ((dx * dx + dy * dy) + dz * dz).astype(np.float32)


def knn_points(X, pc, k, chunk=2048):
    """X [B,Q,3], pc [B,M,3] -> (pts [B,Q,k,3], dists [B,Q,k], idx [B,Q,k] int64)."""
    X = np.asarray(X, np.float32)
    pc = np.asarray(pc, np.float32)
    B, Q, _ = X.shape
    idx = np.empty((B, Q, k), np.int64)
    d2k = np.empty((B, Q, k), np.float32)
    for b in range(B):
        for q0 in range(0, Q, chunk):
            d2 = d2_matrix(X[b, q0:q0 + chunk], pc[b])
            order = np.argsort(d2, axis=1, kind="stable")[:, :k]        # stable: ties -> lower index
            idx[b, q0:q0 + chunk] = order
            d2k[b, q0:q0 + chunk] = np.take_along_axis(d2, order, axis=1)
    dists = np.sqrt(d2k).astype(np.float32)
    pts = np.stack([pc[b][idx[b]] for b in range(B)])                   # knn_gather
    return pts, dists, idx


def knn_offsets(X, pc, k):
    """SconeOcc.py:293-298: neighbours minus the query."""
    pts, dists, idx = knn_points(X, pc, k)
    return (pts - np.asarray(X, np.float32)[:, :, None, :]).astype(np.float32), dists, idx


def tie_aware_index_match(idx_a, d_a, idx_b, d_b, X, pc, rtol=0.0, atol=0.0):
    """True if the two kNN results agree up to permutations inside groups of (near-)equal distance.
    Every index set must be explained: for each query, the multiset of distances must agree within
    (atol, rtol), and any index present in one result but not the other must lie at a distance within
    tolerance of the k-th distance (a boundary tie) or be a within-group permutation."""
    idx_a, idx_b = np.asarray(idx_a), np.asarray(idx_b)
    d_a, d_b = np.asarray(d_a, np.float64), np.asarray(d_b, np.float64)
    if idx_a.shape != idx_b.shape:
        return False
    tol = atol + rtol * np.maximum(np.abs(d_a), np.abs(d_b))
    if not np.all(np.abs(d_a - d_b) <= tol):
        return False
    B, Q, k = idx_a.shape
    X = np.asarray(X, np.float64)
    pc = np.asarray(pc, np.float64)
    for b in range(B):
        diff_rows = np.nonzero(np.any(idx_a[b] != idx_b[b], axis=1))[0]
        for q in diff_rows:
            sa, sb = set(idx_a[b, q].tolist()), set(idx_b[b, q].tolist())
            kth = max(d_a[b, q, -1], d_b[b, q, -1])
            for i in sa ^ sb:                                   # only allowed if tied with the k-th distance
                di = np.sqrt(((X[b, q] - pc[b, i]) ** 2).sum())
                if abs(di - kth) > atol + rtol * kth + 1e-12:
                    return False
            # common indices may be permuted only inside equal-distance groups: check per-position distances
            for j in range(k):
                if idx_a[b, q, j] != idx_b[b, q, j]:
                    da = np.sqrt(((X[b, q] - pc[b, idx_a[b, q, j]]) ** 2).sum())
                    db = np.sqrt(((X[b, q] - pc[b, idx_b[b, q, j]]) ** 2).sum())
                    if abs(da - db) > atol + rtol * max(da, db) + 1e-12:
                        return False
    return True
