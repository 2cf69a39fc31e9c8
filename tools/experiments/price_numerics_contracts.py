"""Price cheaper numerics contracts for the fused local transformer BEFORE writing a kernel (Next #4).

The default kernel (variant 6) carries every matrix operand of the local PCTransformers as fp16 hi/lo planes and spends THREE MFMAs
per fp32 product (x_hi w_hi + x_hi w_lo + x_lo w_hi): measured 5-8e-7 against a 1e-4 bar.  This script emulates, in numpy, what the
occupancies (and the gains behind them) become when every linear layer of `local_transformers.*` runs under a cheaper contract, and
reports the worst error against the fp64 oracle on the reference goldens:

  c6      x_hi w_hi + x_hi w_lo + x_lo w_hi           3 fp16 MFMAs   (today)
  drop_w  x_hi w_hi + x_lo w_hi                       2 fp16 MFMAs   (weights at fp16 precision)
  drop_x  x_hi w_hi + x_hi w_lo                       2 fp16 MFMAs   (activations at fp16 precision)
  fp16    x_hi w_hi                                   1 fp16 MFMA
  fp8x    x_hi w_hi + q8(x_hi) q8(w_lo) + q8(x_lo) q8(w_hi)   1 fp16 MFMA + 1 fp8 MFMA over a doubled K (fp8 runs at 2x the fp16 rate
          on gfx950: the cost of 2 fp16 MFMAs); q8 = e4m3 with a per-tensor power-of-two scale
  bf16x3  x_hi w_hi + x_hi w_mid + x_mid w_hi  (bf16 hi/mid)   3 bf16 MFMAs = today's count: listed for reference only

Accumulation is fp32 everywhere (float32 matmul).  LayerNorm, GELU, softmax, the 16 x 16 attention products and everything outside
the local transformers stay fp32: they are < 3 % of the path's flops.  A contract is worth a kernel only with >= 3x margin to 1e-4.
    python tools/experiments/price_numerics_contracts.py            (CPU, ~2 min)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import weights                                           # noqa: E402
from oracle import nets                                  # noqa: E402


def split16(a):
    hi = a.astype(np.float16).astype(np.float32)
    lo = (a - hi).astype(np.float16).astype(np.float32)
    return hi, lo


def bf16(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)      # round to nearest even
    return r.view(np.float32)


def e4m3(a, per_row=False, block=0):
    """Round to fp8 e4m3 (3 mantissa bits, exponents 2^-6 .. 2^8, subnormals, saturating at 448) after a power-of-two scale: one per
    tensor, per row (per_row: what a kernel can find with one wave reduction per token), or per block of `block` consecutive K
    elements of a row (the E8M0 block scales of gfx950's v_mfma_scale_f32_*_f8f6f4)."""
    a = np.asarray(a, np.float32)
    if block:
        sh = a.shape
        b = a.reshape(-1, sh[-1] // block, block) if sh[-1] % block == 0 else None
        if b is not None:
            m = np.abs(b).max(-1, keepdims=True)
            s = 2.0 ** np.floor(np.log2(448.0 / np.maximum(m, 1e-30)))
            x = b * s
            e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -6)))
            q = 2.0 ** (e - 3)
            return (np.clip(np.round(x / q) * q, -448.0, 448.0) / s).reshape(sh).astype(np.float32)
    m = np.abs(a).max(-1, keepdims=True) if per_row else np.float32(np.abs(a).max())
    if not per_row and float(m) == 0.0:
        return a
    s = 2.0 ** np.floor(np.log2(448.0 / np.maximum(m, 1e-30)))
    x = a * s.astype(np.float32) if per_row else a * np.float32(s)
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -6)))
    q = 2.0 ** (e - 3)
    y = np.clip(np.round(x / q) * q, -448.0, 448.0)
    return (y / s).astype(np.float32)


def e4m3_fixed(a, scale):
    x = np.asarray(a, np.float32) * np.float32(scale)
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -6)))
    q = 2.0 ** (e - 3)
    return (np.clip(np.round(x / q) * q, -448.0, 448.0) / np.float32(scale)).astype(np.float32)


def product(contract, x, w):
    """x [..., K] @ w [N, K]^T under `contract`, fp32 accumulation."""
    x, w = np.asarray(x, np.float32), np.asarray(w, np.float32)
    if contract == "fp32":
        return x @ w.T
    ws = np.float32(2.0 ** np.round(np.log2(1.0 / max(float(np.abs(w).max()), 1e-30))))      # per-matrix power-of-two weight scale
    wsc = w * ws
    if contract == "bf16x3":
        xh = bf16(x); xm = bf16(x - xh); wh = bf16(wsc); wm = bf16(wsc - wh)
        return (xh @ wh.T + xh @ wm.T + xm @ wh.T) / ws
    xh, xl = split16(x)
    wh, wl = split16(wsc)
    if contract == "c6":
        y = xh @ wh.T + xh @ wl.T + xl @ wh.T
    elif contract == "drop_w":
        y = xh @ wh.T + xl @ wh.T
    elif contract == "drop_x":
        y = xh @ wh.T + xh @ wl.T
    elif contract == "fp16":
        y = xh @ wh.T
    elif contract == "fp8x":
        y = xh @ wh.T + (e4m3(xh) @ e4m3(wl).T + e4m3(xl) @ e4m3(wh).T)
    elif contract == "fp8x_row":                     # activations scaled per token (row), weights per output channel (row of W)
        y = xh @ wh.T + (e4m3(xh, True) @ e4m3(wl, True).T + e4m3(xl, True) @ e4m3(wh, True).T)
    elif contract == "fp8x_blk":                     # E8M0 scales per 32 K-elements on both operands (MX block scaling)
        y = xh @ wh.T + (e4m3(xh, block=32) @ e4m3(wl, block=32).T + e4m3(xl, block=32) @ e4m3(wh, block=32).T)
    elif contract == "fp8x_fix":                     # ONE static scale for the activations' hi part (2^0: |x| < 448), lo = hi * 2^-11
        f8 = lambda t, sc: e4m3_fixed(t, sc)
        y = xh @ wh.T + (f8(xh, 1.0) @ e4m3(wl, True).T + f8(xl, 2048.0) @ e4m3(wh, True).T)
    else:
        raise ValueError(contract)
    return y / ws


def run(contract, sd, g, case, dtype=np.float32):
    orig = nets._lin

    def lin(sd_, name, x):
        if name.startswith("local_transformers.") and contract != "fp64":
            return (product(contract, x, sd_[name + ".weight"]) + sd_[name + ".bias"].astype(np.float32)).astype(np.float32)
        return orig(sd_, name, x)
    nets._lin = lin
    try:
        perms = [g[f"{case}_perm{i}"] for i in range(3)]
        return nets.scone_occ_forward(sd, g[f"{case}_pc"], g[f"{case}_x"], g[f"{case}_vh"], perms, dtype=dtype)
    finally:
        nets._lin = orig


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "scone_occ.npz"))
    import contextlib, io, importlib
    M = importlib.import_module("macarons_amd.networks.SconeOcc")
    with contextlib.redirect_stdout(io.StringIO()):
        occ = M.SconeOcc()
    sd = weights.make_state_dict(weights.shapes_of(occ), 2)          # the goldens' weights (tests/golden/make_golden.py: gen_occ, seed 2)
    sd = {k: np.asarray(v) for k, v in sd.items()}
    # the goldens were generated with seeded reference-style weights: check that this state dict reproduces the golden in fp32
    cases = ("m1024_q300", "m4096_q512")
    contracts = ("c6", "bf16x3", "fp8x", "fp8x_row", "fp8x_blk", "fp8x_fix", "drop_w", "drop_x", "fp16")
    print(f"{'contract':8s}  " + "  ".join(f"{c:>22s}" for c in cases) + "      (max |occ - occ_fp64| / max |occ_fp64|; the bar is 1e-4, a kernel needs <= 3.3e-5)")
    ref64 = {}
    for case in cases:
        ref64[case] = run("fp64", sd, g, case, np.float64)
        y32 = run("fp32", sd, g, case)
        gold = g[f"{case}_y"]
        print(f"# {case}: fp32 oracle vs fp64 oracle {np.abs(y32 - ref64[case]).max() / np.abs(ref64[case]).max():.2e}; "
              f"vs the reference's golden {np.abs(y32 - gold).max() / np.abs(gold).max():.2e}  (weights seed check)")
    for c in contracts:
        errs = []
        for case in cases:
            y = run(c, sd, g, case)
            errs.append(np.abs(y - ref64[case]).max() / np.abs(ref64[case]).max())
        print(f"{c:8s}  " + "  ".join(f"{e:22.2e}" for e in errs))
    # other weights: another seed, and the same weights with every linear layer of the local transformers 4x larger (activations of
    # trained networks are not unit-scale: the reference's Kaiming / Xavier init gives |activation| ~ 4 in these layers)
    for label, sd2 in (("seed 7", {k: np.asarray(v) for k, v in weights.make_state_dict(weights.shapes_of(occ), 7).items()}),
                       ("4x local weights", {k: (v * 4 if (k.startswith("local_transformers.") and k.endswith("weight") and v.ndim == 2) else v)
                                              for k, v in sd.items()})):
        print(f"---- {label}")
        r64 = {case: run("fp64", sd2, g, case, np.float64) for case in cases}
        for c in ("c6", "fp8x", "fp8x_row", "fp8x_blk", "fp8x_fix"):
            errs = [np.abs(run(c, sd2, g, case) - r64[case]).max() / np.abs(r64[case]).max() for case in cases]
            print(f"{c:8s}  " + "  ".join(f"{e:22.2e}" for e in errs))


if __name__ == "__main__":
    main()
