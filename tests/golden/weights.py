"""Deterministic weights for the network goldens (our own generator; no reference code).

make_state_dict(shapes, seed): `shapes` maps state_dict keys to shapes (taken from the module under test, whose
keys/shapes equal the reference's).  Values follow the reference's init style (scone_utils.py:399-428:
Xavier-normal for w_q/w_k/w_v, Kaiming-normal elsewhere), with non-trivial LayerNorm affine parameters and
biases so every term of the forward is exercised.  The same dict is loaded into the REAL reference modules by
make_golden.py (strict load_state_dict -> also checks key/shape compatibility) and into macarons_amd modules
by the tests.
"""
import numpy as np


def make_state_dict(shapes, seed):
    rng = np.random.default_rng(seed)
    sd = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        leaf = name.split(".")
        if leaf[-1] == "weight" and len(shape) == 2:
            fan_out, fan_in = shape
            if leaf[-2] in ("w_q", "w_k", "w_v"):
                std = np.sqrt(2.0 / (fan_in + fan_out))
            else:
                std = np.sqrt(2.0 / fan_in)
            v = rng.standard_normal(shape) * std
        elif leaf[-1] == "weight":                       # LayerNorm gamma
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif "norm" in leaf[-2]:                         # LayerNorm beta
            v = 0.05 * rng.standard_normal(shape)
        else:                                            # Linear bias
            v = rng.uniform(-0.1, 0.1, shape)
        sd[name] = v.astype(np.float32)
    return sd


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}
