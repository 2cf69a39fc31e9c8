"""Per-shape timing of ops.linear (run on the GPU box)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get('MCR_DEV_LIB'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_libs', f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd import ops
dev = torch.device("cuda:0")
shapes = [(100000, 512, 1344, 1, 0), (100000, 256, 512, 1, 0), (100000, 512, 256, 1, 0), (100000, 256, 128, 1, 0)] if os.environ.get('HEAD') else [(262144, 128, 128, 0, 0), (262144, 192, 128, 0, 0), (262144, 256, 128, 1, 0), (262144, 128, 256, 0, 1),
          (262144, 125, 3, 1, 0), (262144, 125, 125, 0, 0), (100000, 512, 1344, 1, 0), (100000, 256, 512, 1, 0),
          (2048, 384, 256, 0, 0), (2048, 512, 256, 1, 0), (2048, 256, 512, 0, 1)]
for (M, N, K, gelu, res) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    for _ in range(3): ops.linear(x, w, b, gelu=bool(gelu), residual=r)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 10
    for _ in range(n): ops.linear(x, w, b, gelu=bool(gelu), residual=r)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    fl = 2.0 * M * N * K; by = 4.0 * (M * K + M * N * (2 if res else 1) + N * K)
    ref = torch.nn.functional.linear(x[:4096].double(), w.double(), b.double())
    if gelu: ref = torch.nn.functional.gelu(ref)
    if res: ref = ref + r[:4096].double()
    y = ops.linear(x, w, b, gelu=bool(gelu), residual=r)[:4096].double()
    err = float((y - ref).abs().max() / ref.abs().max())
    print(f"[MCR_LINEAR3={os.environ.get('MCR_LINEAR3','1')}] err {err:.1e} M={M:7d} N={N:4d} K={K:5d} gelu={gelu} res={res}: {dt*1e6:9.1f} us  {fl/dt/1e12:6.1f} TFLOP/s  {by/dt/1e9:7.0f} GB/s")
