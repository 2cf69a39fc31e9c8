"""Dev: the long-sequence attention kernel at the batch sizes of a MACARONS decision (30 clouds x 2048 tokens, heads (16, 64): SconeVis;
41 x 2048, heads (8, 32): the global PCTransformer of the occupancy field).  MCR_DEV_LIB selects an experimental library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib
if os.environ.get("MCR_DEV_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{os.environ['MCR_DEV_LIB']}.so")
from macarons_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = int(os.environ.get("REPS", 20))
for (S, L, H, qk, v) in [(30, 2048, 4, 64, 256), (41, 2048, 4, 32, 128), (8, 2048, 4, 64, 256), (1, 2048, 4, 64, 256)]:
    qkv = torch.randn(S, L, 2 * qk + v, device=dev)
    for _ in range(3): y = ops.attention_packed(qkv, H, qk, v)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): y = ops.attention_packed(qkv, H, qk, v)
    e1.record(); torch.cuda.synchronize()
    flop = 2.0 * S * H * L * L * (qk // H + v // H)
    us = e0.elapsed_time(e1) / N * 1e3
    print(f"[{os.environ.get('MCR_DEV_LIB', 'main')}] S={S} L={L} dq={qk//H} dv={v//H}: {us:.1f} us  {flop / us * 1e-6:.1f} TF/s algorithmic  "
          f"checksum {float(y.double().abs().sum()):.6f}")
