"""`torch.ops.macarons.*` — the operator names SURVEY §8(b) lists for the layer below the Python class boundary: a C++
`TORCH_LIBRARY` extension (macarons_amd/csrc_torch/macarons_torch.cpp -> libmacarons_torch.so, built by macarons_amd.build):
`at::Tensor` in / out, launches on `c10::hip::getCurrentHIPStream()`, errors through `TORCH_CHECK`; each operator is a shim over
the C ABI of `include/macarons_hip.h`.  Registered on the CUDA (= HIP) dispatch key only -- calling one with CPU tensors raises
torch's own "no kernel for backend CPU" error; the MI355X path has no fallback.  Importing this module loads the library:

    import macarons_amd.torch_ops                      # once
    gains = torch.ops.macarons.sh_coverage_gain(pts, harmonics, cams, True)

Tensor lists (`weights`, `pc_scales`, `local_blobs`) are what `SconeVis.weight_table()` / `SconeOcc.weight_table()` /
`networks.packing.pack_local_pct` return.  All operators launch on the current HIP stream and never synchronise, except
`sample_proxy`, whose unique count sizes its outputs (the reference's `torch.unique` synchronises at the same point).
"""
import os

import torch

from . import _lib

LIB_PATH = os.path.join(_lib.PKG_DIR, "libmacarons_torch.so")
_NAMES = ("sh_coverage_gain", "sh_coverage_gain_best", "sh_visibilities", "knn_gather_offset", "points_in_fov", "view_state", "view_harmonics", "sample_proxy",
          "scone_vis_forward", "scone_occ_forward")

if not os.path.exists(LIB_PATH):
    raise _lib.MacaronsHipError(f"{LIB_PATH} not found: build it with `python -m macarons_amd.build` "
                                "(torch.ops.macarons.* is the C++ TORCH_LIBRARY extension; there is no Python fallback)")
_lib.lib()                                # libmacarons_hip.so first (torch is imported above: one HIP runtime in the process)
torch.ops.load_library(LIB_PATH)


def registered():
    """Names of the operators under torch.ops.macarons."""
    return sorted(_NAMES)
