"""Host-side weight packing of the fused local transformers (no GPU): blob sizes match what the kernels expect."""
import io
import contextlib

import numpy as np
import torch


def _pct():
    from macarons_amd.networks import SconeOcc
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        occ = SconeOcc()
    return occ.local_transformers[0]


def test_blob_sizes_match_the_kernels():
    from macarons_amd import _lib
    from macarons_amd.networks.packing import pack_local_pct
    pct = _pct()
    L = _lib.lib()
    assert pack_local_pct(pct, 1).numel() == L.mcr_local_pct_blob_floats()
    assert pack_local_pct(pct, 5).numel() == L.mcr_local_pct3_blob_floats()
    assert pack_local_pct(pct, 6).numel() == L.mcr_local_pct6_blob_floats()
