"""Dev: randomised equality of the grid-pruned kNN (mcr_knn_points_grid) and the brute-force kernels (mcr_knn_points): sizes, shapes of
the clouds, scales, ties.  python tools/fuzz_knn_grid.py [n_cases] [seed]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops
from macarons_amd._lib import lib, check, c_i64, c_int

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ptr = lambda t: ctypes.c_void_p(t.data_ptr())


def cloud(kind, n, scale, shift):
    if kind == "uniform":
        p = rng.uniform(-.5, .5, (n, 3))
    elif kind == "clusters":
        c = rng.uniform(-.5, .5, (rng.integers(1, 6), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, rng.choice([1e-4, 1e-2, 0.1]), (n, 3))
    elif kind == "shell":
        d = rng.normal(size=(n, 3)); p = 0.3 * d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.5, 1.5, 3)
    elif kind == "planar":
        p = rng.uniform(-.5, .5, (n, 3)); p[:, rng.integers(0, 3)] = rng.uniform(-.5, .5)
    elif kind == "lattice":
        p = rng.integers(0, rng.integers(3, 12), (n, 3)) / 8.0
    else:
        p = rng.uniform(-.5, .5, (n, 3)); p[: n // 2] = p[n // 2: 2 * (n // 2)]          # exact duplicates
    return (p * scale + shift).astype(np.float32)


bad = 0
for case in range(n_cases):
    B = int(rng.choice([1, 1, 2]))
    M = int(rng.integers(1024, 16385)); Q = int(rng.choice([1, 31, 33, rng.integers(100, 6000)]))
    scale = float(rng.choice([1e-3, 1.0, 1.0, 50.0])); shift = float(rng.choice([0.0, 0.0, 10.0, -300.0])) * scale
    kp, kq = rng.choice(["uniform", "clusters", "shell", "planar", "lattice", "dups"]), rng.choice(["uniform", "clusters", "shell", "lattice"])
    pc = torch.from_numpy(np.stack([cloud(kp, M, scale, shift) for _ in range(B)])).to(dev)
    X = torch.from_numpy(np.stack([cloud(kq, Q, scale * rng.choice([0.05, 1.0, 3.0]), shift) for _ in range(B)])).to(dev)
    p1, d1, i1 = ops.knn_points(X, pc, 16, True)
    i0 = torch.empty_like(i1); d0 = torch.empty_like(d1); p0 = torch.empty_like(p1)
    check(lib().mcr_knn_points(ptr(X), ptr(pc), ptr(i0), ptr(d0), ptr(p0), c_i64(B), c_i64(Q), c_i64(M), c_int(16), c_int(1),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mcr_knn_points")
    torch.cuda.synchronize()
    ok = torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(p0, p1)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: B={B} M={M} Q={Q} cloud={kp} queries={kq} scale={scale} shift={shift}: "
              f"{int((i0 != i1).any(-1).sum())} queries differ")
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
