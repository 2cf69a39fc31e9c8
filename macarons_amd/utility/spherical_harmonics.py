"""Real spherical harmonics table builder (setup-time only, plain torch on any device).

Same convention as macarons/utility/spherical_harmonics.py:67-157 (Condon-Shortley phase, channel k = l*l+l+m).
Used once per run to build the [64, 98] base-harmonics table of the view-state grid; the per-pair SH
evaluation of the hot path lives in csrc/sh_scorer.hip.
"""
import math

import torch


def _semifactorial(x):
    r = 1.0
    for v in range(x, 1, -2):
        r *= v
    return r


def _norm(l, m):
    n = math.sqrt((2 * l + 1) / (4 * math.pi))
    if m:
        n *= math.sqrt(2.0 * math.factorial(l - m) / math.factorial(l + m))
    return n


def clear_spherical_harmonics_cache():
    """API compatibility with the reference (spherical_harmonics.py:33): nothing is cached here."""


def get_spherical_harmonics(l, theta, phi):
    """-> [*theta.shape, 2l+1], m = -l..l  (spherical_harmonics.py:143-157)."""
    x = torch.cos(theta)
    P = {}
    for m in range(l + 1):
        P[(m, m)] = torch.ones_like(x) if m == 0 else ((-1) ** m * _semifactorial(2 * m - 1)) * torch.pow(1 - x * x, m / 2)
        for ll in range(m + 1, l + 1):
            y = ((2 * ll - 1) / (ll - m)) * x * P[(ll - 1, m)]
            if ll - m > 1:
                y = y - ((ll + m - 1) / (ll - m)) * P[(ll - 2, m)]
            P[(ll, m)] = y
    out = []
    for m in range(-l, l + 1):
        a = abs(m)
        if m == 0:
            out.append(_norm(l, 0) * P[(l, 0)])
        else:
            ang = torch.cos(m * phi) if m > 0 else torch.sin(a * phi)
            out.append(ang * P[(l, a)] * _norm(l, a))
    return torch.stack(out, dim=-1)
