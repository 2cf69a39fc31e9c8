"""ctypes loader for the plain-C oracle (oracle/csrc/*.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle_port.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n):
    """OpenMP threads of the following calls (0: OpenMP's default)."""
    lib().scorer_port_set_threads(ctypes.c_int(int(n)))


def coverage_gain(pts, harmonics, cams, use_sigmoid=True):
    """C port of SconeVis.compute_coverage_gain. Returns (gains [B,C], n_threads)."""
    pts, harmonics, cams = _f(pts), _f(harmonics), _f(cams)
    B, N, P = pts.shape
    C = cams.shape[1]
    out = np.empty((B, C), np.float32)
    fn = lib().scorer_port_coverage_gain
    fn.restype = ctypes.c_int
    nt = fn(_ptr(pts), ctypes.c_int(P), _ptr(harmonics), _ptr(cams), _ptr(out), ctypes.c_int64(B), ctypes.c_int64(N),
            ctypes.c_int64(C), ctypes.c_int(int(use_sigmoid)))
    return out, nt


def visibilities(pts, harmonics, cams, use_sigmoid=True):
    pts, harmonics, cams = _f(pts), _f(harmonics), _f(cams)
    B, N, P = pts.shape
    C = cams.shape[1]
    out = np.empty((B, C, N), np.float32)
    fn = lib().scorer_port_visibilities
    fn.restype = ctypes.c_int
    fn(_ptr(pts), ctypes.c_int(P), _ptr(harmonics), _ptr(cams), _ptr(out), ctypes.c_int64(B), ctypes.c_int64(N),
       ctypes.c_int64(C), ctypes.c_int(int(use_sigmoid)))
    return out
