"""Host-side weight packing of the fused local transformers (no GPU): blob sizes match what the kernels expect."""
import io
import os
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
    assert pack_local_pct(pct, 7).numel() == L.mcr_local_pct7_blob_floats()


def test_variant_7_blob_is_the_unscaled_fp16_weights_in_fragment_order():
    """The opt-in 16-bit matrix path (local_pct7.hip): matrices as ONE fp16 plane in the MFMA-fragment order [n-tile][k16][lane][8] --
    fp16(W) without variant 6's per-matrix power of two -- followed by the UNSCALED bias vectors and nothing else."""
    from macarons_amd.networks.packing import pack_local_pct
    pct = _pct()
    b7, b6 = pack_local_pct(pct, 7), pack_local_pct(pct, 6)
    n_vec = 2 * 128 + 2 * (192 + 128 + 256 + 128) + 128
    n_mat7 = b7.numel() - n_vec
    assert 2 * n_mat7 == b6.numel() - n_vec - 32                          # half a float per weight against variant 6's one
    # matrix 14 (the last one): linear0 with the final LayerNorm's gamma folded, rows = features, 128 x 128
    with torch.no_grad():
        W = (pct.linear0.weight * pct.norm.weight[None, :]).float()
    halves = b7[:n_mat7].view(torch.float16)
    frag = halves[-128 * 128:].view(4, 8, 2, 32, 8)                        # [nt][k16 step][lane half h][lane j][8]
    nt, s_, h, j = 2, 5, 1, 17
    assert torch.equal(frag[nt, s_, h, j], W[32 * nt + j, 16 * s_ + 8 * h:16 * s_ + 8 * h + 8].to(torch.float16))
    # the bias of linear0 (+ W @ beta of the folded LayerNorm) closes the blob, unscaled
    with torch.no_grad():
        c = pct.linear0.bias + pct.linear0.weight @ pct.norm.bias
    assert torch.allclose(b7[-128:], c.float(), rtol=0, atol=1e-6)


def test_param_fingerprint_sees_every_kind_of_weight_change():
    """The cache key of the derived weight images must change on: optimizer-style in-place updates, `p.data = new`,
    load_state_dict(assign=True), parameter replacement -- on ANY parameter, not just the first / last."""
    from macarons_amd.networks.packing import _param_key, invalidate
    pct = _pct()
    cache = {}
    k0 = _param_key(pct, cache)
    assert _param_key(pct, cache) == k0                                   # stable when nothing changed
    mid = pct.encoders[0].ff.linear1                                      # a parameter in the middle of the list
    with torch.no_grad():
        mid.weight.add_(1.0)
    k1 = _param_key(pct, cache)
    assert k1 != k0
    mid.weight.data = mid.weight.data.clone()
    k2 = _param_key(pct, cache)
    assert k2 != k1
    mid.weight = torch.nn.Parameter(mid.weight.detach().clone())
    k3 = _param_key(pct, cache)
    assert k3 != k2
    sd = {k: v.clone() for k, v in pct.state_dict().items()}
    pct.load_state_dict(sd, assign=True)
    k4 = _param_key(pct, cache)
    assert k4 != k3
    pct.load_state_dict(sd)                                               # copy_ in place: versions bump
    k5 = _param_key(pct, cache)
    assert k5 != k4
    invalidate(cache)
    assert _param_key(pct, cache) != k5


def test_h2d_is_a_plain_copy_off_the_gpu():
    """ops.h2d (the stall-free upload of the MACARONS glue: pinned staging + asynchronous copy on a GPU) degrades to a plain,
    dtype-converting .to() on a CPU target."""
    import numpy as np
    import torch
    from macarons_amd import ops
    t = ops.h2d(np.arange(5), torch.int32, "cpu")
    assert t.dtype == torch.int32 and t.tolist() == [0, 1, 2, 3, 4]
    t = ops.h2d([1.5, 2.5], torch.float32, torch.device("cpu"))
    assert t.dtype == torch.float32 and t.tolist() == [1.5, 2.5]


def test_frozen_fingerprint_is_opt_in_and_reversible():
    """freeze(): the key is taken once and returned unchanged whatever happens to the parameters; invalidate() / freeze(False) go
    back to fingerprinting every call."""
    import torch
    from macarons_amd.networks.packing import _param_key, freeze, invalidate
    m = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
    c = {}
    k0 = _param_key(m, c)
    freeze(m, c)
    with torch.no_grad():
        m[0].weight.add_(1.0)
    assert _param_key(m, c) == k0                       # frozen: the edit goes unnoticed (documented)
    freeze(m, c, False)
    k1 = _param_key(m, c)
    assert k1 != k0
    freeze(m, c)
    invalidate(c)
    with torch.no_grad():
        m[1].bias.add_(1.0)
    assert _param_key(m, c) != k1


def test_host_thread_cap_follows_the_quota_and_never_raises_the_count(monkeypatch):
    from macarons_amd.utility import host
    n = host.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    try:
        monkeypatch.setenv("MCR_HOST_THREADS", "0")
        assert host.limit_host_threads() == before                 # opt-out
        monkeypatch.setenv("MCR_HOST_THREADS", str(before + 7))
        assert host.limit_host_threads() == before                 # a cap, not a request for more
        monkeypatch.delenv("MCR_HOST_THREADS")
        assert host.limit_host_threads() <= max(n, 1) or before <= n
        assert host.limit_host_threads(1) == 1
    finally:
        torch.set_num_threads(before)
