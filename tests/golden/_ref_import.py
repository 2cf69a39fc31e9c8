"""Import the *reference* MACARONS networks from /root/reference (this container only).

Used ONLY by tests/golden/make_golden.py to produce the committed golden vectors.
The reference is pure Python; it needs `pytorch3d` (absent here).  On the hot path
only `pytorch3d.ops.knn_gather` has behaviour (a pure index gather), so we install
MagicMock stubs for the pytorch3d/torchvision/matplotlib module tree and a real
3-line `knn_gather`.  Nothing from the reference is copied into this repo, and
`sys.dont_write_bytecode` keeps the read-only mount clean.
"""
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = "/root/reference"


def _knn_gather(x, idx, lengths=None):
    # x [B,M,U], idx [B,L,K] -> [B,L,K,U]   (behaviour of pytorch3d.ops.knn_gather)
    import torch
    B, M, U = x.shape
    _, L, K = idx.shape
    flat = idx.reshape(B, L * K, 1).expand(-1, -1, U)
    return torch.gather(x, 1, flat).reshape(B, L, K, U)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


_STUB_ROOTS = ("pytorch3d", "torchvision", "matplotlib", "mpl_toolkits", "plotly", "skimage", "PIL",
               "imageio", "cv2", "gradio", "open3d")


class _StubFinder:
    """Meta-path finder: any (sub)module under _STUB_ROOTS that is not installed becomes a stub package."""

    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        if module.__name__ == "pytorch3d.ops":
            module.knn_gather = _knn_gather


def install_stubs():
    sys.dont_write_bytecode = True
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())   # appended: real installs (if any) win
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference():
    install_stubs()
    import importlib
    mods = {}
    for n in ["macarons.networks.SconeVis", "macarons.networks.SconeOcc", "macarons.networks.Attention",
              "macarons.utility.spherical_harmonics", "macarons.utility.CustomGeometry",
              "macarons.utility.utils"]:
        mods[n.split(".")[-1]] = importlib.import_module(n)
    return mods
