"""macarons_amd — MI355X (gfx950)-native SCONE coverage-gain hot path of MACARONS.

Host side mirrors the reference's `macarons.networks` class surface; the work is done by hand-written
HIP kernels in libmacarons_hip.so behind the C ABI of include/macarons_hip.h.
"""
__version__ = "0.1.0"
from .patch import patch_reference, unpatch_reference   # noqa: E402,F401  (the reference-side binding)
