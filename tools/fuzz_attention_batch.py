"""Dev fuzz: the long-sequence attention on random batches (sequence count, length, head shape, a V tile outside the fp16 range now and
then) == every sequence alone, bit for bit without the key split and within rounding with it; MCR_ATTN_QG2=0 must give the same bits."""
import os, sys, subprocess
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
n_bad = 0
for it in range(int(os.environ.get("N", 40))):
    H = 4
    qk, v = ((64, 256), (32, 128))[int(rng.integers(2))]
    S, L = int(rng.integers(4, 44)), int(rng.integers(200, 2400))
    qkv = torch.from_numpy(rng.standard_normal((S, L, 2 * qk + v)).astype(np.float32)).to(dev)
    if rng.random() < 0.3:
        b, t = int(rng.integers(S)), int(rng.integers(L))
        qkv[b, t:t + 7, 2 * qk:] *= 1e6
    big = ops.attention_packed(qkv, H, qk, v, split=False)
    big_split = ops.attention_packed(qkv, H, qk, v)           # small batches of long sequences split their keys (the kernel's predicate)
    for b in {0, S - 1, int(rng.integers(S))}:
        one = ops.attention_packed(qkv[b:b + 1].contiguous(), H, qk, v, split=False)
        if not torch.equal(one[0], big[b]):
            n_bad += 1
            print("MISMATCH", it, (S, L, qk, v), b, float((one[0] - big[b]).abs().max()))
        err = float(((one[0] - big_split[b]).abs() / one[0].abs().clamp(min=1.0)).max())
        if not err <= 5e-6:                                   # the key split: within rounding of the unsplit result
            n_bad += 1
            print("SPLIT MISMATCH", it, (S, L, qk, v), b, err)
    if not torch.isfinite(big).all():
        n_bad += 1; print("non-finite", it)
print("fuzz_attention_batch:", "OK" if n_bad == 0 else f"{n_bad} FAILURES")
