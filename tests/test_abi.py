"""The C-ABI library loads and exports every symbol include/macarons_hip.h declares (no GPU needed)."""
import ctypes
import os
import subprocess

import pytest

from macarons_amd import _lib, build


@pytest.fixture(scope="module")
def libpath():
    return build.build()


def test_library_builds_and_loads(libpath):
    assert os.path.exists(libpath)
    L = _lib.lib()
    assert L.mcr_abi_version() >= 1
    assert L.mcr_target_arch() == b"gfx950"


def test_every_declared_symbol_is_exported(libpath):
    names = _lib.declared_symbols()
    assert len(names) >= 5
    L = ctypes.CDLL(libpath)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/macarons_hip.h but not exported: {missing}"


def test_exports_are_declared(libpath):
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T mcr_" in l}
    undeclared = exported - set(_lib.declared_symbols())
    assert not undeclared, f"exported but missing from include/macarons_hip.h: {sorted(undeclared)}"


def test_code_object_is_gfx950(libpath):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", libpath], capture_output=True, text=True)
    txt = out.stdout + out.stderr
    assert "gfx950" in txt


def test_ops_refuse_cpu_tensors():
    import torch
    from macarons_amd import ops
    with pytest.raises(_lib.MacaronsHipError):
        ops.sh_coverage_gain(torch.zeros(1, 4, 4), torch.zeros(1, 4, 64), torch.zeros(1, 2, 3))


def test_torch_ops_registered_without_cpu_fallback():
    """torch.ops.macarons.* (SURVEY 8b's operator list) exists after importing macarons_amd.torch_ops and has no CPU kernel."""
    import pytest
    import torch
    import macarons_amd.torch_ops as t
    assert t.registered() == sorted(["sh_coverage_gain", "sh_coverage_gain_best", "sh_visibilities", "knn_gather_offset", "points_in_fov", "view_state",
                                     "view_harmonics", "sample_proxy", "scone_vis_forward", "scone_occ_forward"])
    for name in t.registered():
        assert hasattr(torch.ops.macarons, name)
    with pytest.raises(NotImplementedError):
        torch.ops.macarons.sh_coverage_gain(torch.zeros(1, 4, 3), torch.zeros(1, 4, 64), torch.zeros(1, 2, 3), True)


def test_header_is_valid_c():
    """include/macarons_hip.h is what a non-Python host compiles against (and what libmacarons_torch.so is built with): it must parse as
    plain C -- a comment left open once broke the extension's build without any Python test noticing."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-Wall", os.path.join(root, "include", "macarons_hip.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
