"""ctypes binding of libmacarons_hip.so (declared in include/macarons_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, we raise.
"""
import ctypes
import os
import re

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmacarons_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(PKG_DIR), "include", "macarons_hip.h")


class MacaronsHipError(RuntimeError):
    pass


_lib = None

c_f32p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_size = ctypes.c_size_t
c_vp = ctypes.c_void_p
c_f32 = ctypes.c_float


def declared_symbols(header_path=HEADER_PATH):
    """Every function name declared in include/macarons_hip.h."""
    with open(header_path) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mcr_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        # torch first: it ships its own libamdhip64; loading ours before it would bring up a second HIP
        # runtime (/opt/rocm) in the process and every launch would fail with "no ROCm-capable device".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise MacaronsHipError(
                f"{LIB_PATH} not found: build it with `python -m macarons_amd.build` "
                "(there is no CPU fallback for the MI355X hot path)")
        L = ctypes.CDLL(LIB_PATH)
        L.mcr_last_error.restype = ctypes.c_char_p
        L.mcr_target_arch.restype = ctypes.c_char_p
        for name in declared_symbols():
            fn = getattr(L, name)      # raises AttributeError if the .so lacks a declared symbol
            if name.endswith("_workspace_bytes"):
                fn.restype = c_size
            elif name not in ("mcr_last_error", "mcr_target_arch"):
                fn.restype = c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise MacaronsHipError(f"{what} failed (rc={rc}): {lib().mcr_last_error().decode()}")
