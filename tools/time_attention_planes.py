"""Dev: the planes attention (attention_planes.hip) at the batch sizes of a MACARONS decision and of a single cloud, keys split / unsplit,
next to the fp32-operand kernel.  The timed region includes the split of the packed rows into planes (one elementwise pass the encoders
do not have: their QKV projection writes planes), reported separately by timing it alone."""
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


def bench(f):
    for _ in range(3): y = f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): y = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3, y


for (S, L, H, qk, v) in [(30, 2048, 4, 64, 256), (41, 2048, 4, 32, 128), (8, 2048, 4, 64, 256), (1, 2048, 4, 64, 256), (1, 2048, 4, 32, 128)]:
    qkv = torch.randn(S, L, 2 * qk + v, device=dev)
    us0, y0 = bench(lambda: ops.attention_packed(qkv, H, qk, v))
    line = f"S={S} L={L} dq={qk//H} dv={v//H}: fp32 operands {us0:.1f} us"
    for mode in (1, 0):
        us, y = bench(lambda: ops.attention_packed_planes(qkv, H, qk, v, split_mode=mode))
        line += f" | planes split_mode={mode}: {us:.1f} us (max diff {float((y - y0).abs().max()):.2e})"
    print(f"[{os.environ.get('MCR_DEV_LIB', 'main')}] " + line)
