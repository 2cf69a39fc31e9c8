"""Dev: long-sequence attention kernel timing, one block per (query tile, head) vs keys split over two blocks
(MCR_ATTN_MFMA=0 selects the VALU reference kernel)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops
dev = torch.device("cuda:0")
for (S, L, H, qk, v) in [(1, 2048, 4, 64, 256), (1, 2048, 4, 32, 128), (1, 1777, 4, 64, 256), (3, 333, 4, 32, 128)]:
    qkv = torch.randn(S, L, 2 * qk + v, device=dev)
    q, k, vv = qkv[..., :qk].double(), qkv[..., qk:2 * qk].double(), qkv[..., 2 * qk:].double()
    hs = lambda t, d: t.reshape(S, L, H, d).transpose(1, 2)
    ref = (torch.softmax(hs(q, qk // H) @ hs(k, qk // H).transpose(-1, -2) / (qk // H) ** 0.5, -1) @ hs(vv, v // H)).transpose(1, 2).reshape(S, L, v)
    for split in (False, True):
        for _ in range(5): y = ops.attention_packed(qkv, H, qk, v, split=split)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): y = ops.attention_packed(qkv, H, qk, v, split=split)
        e1.record(); torch.cuda.synchronize()
        print(f"[MCR_ATTN_MFMA={os.environ.get('MCR_ATTN_MFMA','1')} key-split scratch={split}] S={S} L={L} dq={qk//H} dv={v//H}: "
              f"{e0.elapsed_time(e1)/50*1e3:.1f} us  err {float((y.double()-ref).abs().max()):.1e}")
