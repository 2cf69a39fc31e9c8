"""Dev: the gain kernel ALONE (mcr_sh_coverage_gain_partials) and the gains of the product library against an experimental one
(tools/_libs/libmacarons_hip_<MCR_DEV_LIB>.so, both loaded in one process): us per launch of each, worst relative gain difference."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import _lib, ops
import bench
dev = torch.device("cuda:0")
pts, harm, cams = bench.make_inputs(100_000, 200, 1234, dev, cam_offset=0, n_cam_total=200)
res = {}
for name in ["main", os.environ.get("MCR_DEV_LIB", "pk"), "main"]:
    if name != "main":
        _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", f"libmacarons_hip_{name}.so")
    else:
        _lib.LIB_PATH = os.path.join(_lib.PKG_DIR, "libmacarons_hip.so")
    _lib._LIB = None if hasattr(_lib, "_LIB") else None
    for attr in ("_LIB", "_lib", "_handle"):
        if hasattr(_lib, attr):
            setattr(_lib, attr, None)
    for _ in range(200): ops.sh_coverage_gain_partials(pts, harm, cams, True, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): ops.sh_coverage_gain_partials(pts, harm, cams, True, 0)
    e1.record(); torch.cuda.synchronize()
    g = ops.sh_coverage_gain(pts, harm, cams, True, 0).double().cpu()
    res.setdefault(name, g)
    print(f"[{name}] {_lib.LIB_PATH}: kernel {e0.elapsed_time(e1):.2f} us/launch, max rel diff vs main {float(((g - res['main']).abs() / res['main'].abs()).max()):.3e}")
