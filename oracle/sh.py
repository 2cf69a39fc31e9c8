"""Real (tesseral) spherical harmonics + the reference's spherical-coordinate convention.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in numpy:
  * macarons/utility/CustomGeometry.py:27-45   get_spherical_coords
  * macarons/utility/CustomGeometry.py:5-24    get_cartesian_coords
  * macarons/utility/spherical_harmonics.py:67-108  lpmv (associated Legendre, Condon-Shortley)
  * macarons/utility/spherical_harmonics.py:111-140 get_spherical_harmonics_element
  * macarons/utility/spherical_harmonics.py:143-157 get_spherical_harmonics

Two evaluation styles are provided:
  * ``literal`` — the reference's op sequence (asin/cos/acos, cos(m*phi), pow) in the given dtype;
    used to pin the oracle to the reference goldens.
  * ``trigfree`` — the algebraically identical direction-cosine form (what the HIP kernel uses);
    in fp64 it is the accuracy yard-stick for per-point visibilities (SURVEY §7 "ill-conditioned trig").
"""
import math
import numpy as np

MAX_RANK = 8          # l = 0..7  -> 64 harmonics
N_HARMONICS = MAX_RANK * MAX_RANK


def semifactorial(x):
    # spherical_harmonics.py:49-50
    r = 1.0
    for v in range(x, 1, -2):
        r *= v
    return r


def pochhammer(x, k):
    # spherical_harmonics.py:53-54   x (x+1) ... (x+k-1)
    r = float(x)
    for v in range(x + 1, x + k):
        r *= v
    return r


def spherical_coords(X):
    """CustomGeometry.py:27-45.  X [K,3] -> (r, elev, azim); Y-up, azimuth from +Z toward +X."""
    X = np.asarray(X)
    dt = X.dtype.type
    r = np.sqrt((X * X).sum(axis=1, dtype=X.dtype)).astype(X.dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        yr = X[:, 1] / r
        elev = np.arcsin(yr).astype(X.dtype)
        elev[yr <= -1] = dt(-np.pi / 2)
        elev[yr >= 1] = dt(np.pi / 2)
        q = X[:, 2] / (r * np.cos(elev))
        azim = np.arccos(q).astype(X.dtype)
        azim[q <= -1] = dt(np.pi)
        azim[q >= 1] = dt(0.0)
        azim[X[:, 0] < 0] *= dt(-1)
    return r, elev, azim


def cartesian_coords(r, elev, azim, in_degrees=False):
    """CustomGeometry.py:5-24 (inputs [K,1] or [K]) -> [K,3]."""
    f = np.pi / 180.0 if in_degrees else 1.0
    elev = np.asarray(elev).reshape(-1)
    azim = np.asarray(azim).reshape(-1)
    r = np.asarray(r).reshape(-1, 1)
    X = np.stack((np.cos(f * elev) * np.sin(f * azim), np.sin(f * elev), np.cos(f * elev) * np.cos(f * azim)), axis=1)
    return (r * X).astype(elev.dtype)


def legendre_table(x, dtype):
    """lpmv for all 0<=m<=l<MAX_RANK (spherical_harmonics.py:67-108). Returns dict[(l,m)] -> array."""
    x = np.asarray(x, dtype=dtype)
    P = {}
    one_minus = (1 - x * x).astype(dtype)
    for m in range(MAX_RANK):
        if m == 0:
            P[(0, 0)] = np.ones_like(x)
        else:
            y = (-1) ** m * semifactorial(2 * m - 1)
            P[(m, m)] = (dtype(y) * np.power(one_minus, dtype(m / 2))).astype(dtype)
        for l in range(m + 1, MAX_RANK):
            y = dtype((2 * l - 1) / (l - m)) * x * P[(l - 1, m)]
            if l - m > 1:
                y = y - dtype((l + m - 1) / (l - m)) * P[(l - 2, m)]
            P[(l, m)] = y.astype(dtype)
    return P


def sh_norm(l, m_abs):
    # spherical_harmonics.py:126,138
    N = math.sqrt((2 * l + 1) / (4 * math.pi))
    if m_abs:
        N *= math.sqrt(2.0 / pochhammer(l - m_abs + 1, 2 * m_abs))
    return N


def sh_basis_literal(theta, phi, dtype=np.float32):
    """get_spherical_harmonics for l=0..7, concatenated l-major, m=-l..l  -> [K,64]
    (SconeVis.py:235-239 builds exactly this concatenation)."""
    theta = np.asarray(theta, dtype=dtype)
    phi = np.asarray(phi, dtype=dtype)
    P = legendre_table(np.cos(theta), dtype)
    out = np.empty(theta.shape + (N_HARMONICS,), dtype=dtype)
    for l in range(MAX_RANK):
        for m in range(-l, l + 1):
            a = abs(m)
            if m == 0:
                y = dtype(sh_norm(l, 0)) * P[(l, 0)]
            else:
                y = np.cos(dtype(m) * phi) if m > 0 else np.sin(dtype(a) * phi)
                y = y * P[(l, a)]
                y = y * dtype(sh_norm(l, a))
            out[..., l * l + l + m] = y
    return out


def sh_basis_trigfree(rays, dtype=np.float64):
    """Same basis from direction cosines (no inverse trig): rays [K,3] -> [K,64].
    cos(theta)=y/r, sin(theta)=rho/r, cos(phi)=z/rho, sin(phi)=x/rho  (SURVEY §8 a5)."""
    rays = np.asarray(rays, dtype=dtype)
    x, y, z = rays[:, 0], rays[:, 1], rays[:, 2]
    rho2 = x * x + z * z
    r = np.sqrt(rho2 + y * y)
    rho = np.sqrt(rho2)
    ct = y / r
    st = rho / r
    with np.errstate(invalid="ignore", divide="ignore"):
        cphi = np.where(rho > 0, z / rho, 1.0).astype(dtype)
        sphi = np.where(rho > 0, x / rho, 0.0).astype(dtype)
    # cos(m phi), sin(m phi) by the angle-addition recurrence
    cm = [np.ones_like(cphi), cphi]
    sm = [np.zeros_like(sphi), sphi]
    for m in range(2, MAX_RANK):
        cm.append(cm[m - 1] * cphi - sm[m - 1] * sphi)
        sm.append(sm[m - 1] * cphi + cm[m - 1] * sphi)
    # Legendre with sin(theta)^m instead of pow(1-x^2, m/2)
    P = {}
    for m in range(MAX_RANK):
        if m == 0:
            P[(0, 0)] = np.ones_like(ct)
        else:
            P[(m, m)] = ((-1) ** m * semifactorial(2 * m - 1)) * st ** m
        for l in range(m + 1, MAX_RANK):
            v = ((2 * l - 1) / (l - m)) * ct * P[(l - 1, m)]
            if l - m > 1:
                v = v - ((l + m - 1) / (l - m)) * P[(l - 2, m)]
            P[(l, m)] = v
    out = np.empty((rays.shape[0], N_HARMONICS), dtype=dtype)
    for l in range(MAX_RANK):
        for m in range(-l, l + 1):
            a = abs(m)
            if m == 0:
                v = sh_norm(l, 0) * P[(l, 0)]
            elif m > 0:
                v = sh_norm(l, a) * P[(l, a)] * cm[a]
            else:
                v = sh_norm(l, a) * P[(l, a)] * sm[a]
            out[:, l * l + l + m] = v
    return out
