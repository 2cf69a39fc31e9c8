"""`patch_reference()` -- the reference-side binding of the MI355X path, as one call.

MACARONS has no plugin registry or FFI: its seam for this path is the import graph
    macarons/utility/scone_utils.py:13-14    from ..networks.SconeOcc import SconeOcc
                                             from ..networks.SconeVis import SconeVis, KLDivCE, L1_loss, Uncentered_L1_loss
    macarons/networks/Macarons.py:5-6        from .SconeOcc import * ; from .SconeVis import *
    macarons/networks/SconeVis.py:1, SconeOcc.py:3    from .Attention import *
so the binding installs the `macarons_amd.networks` MODULE objects under the reference's module names and rebinds every name a
reference module imported before the call.  No file of the reference is edited:

    import macarons_amd
    macarons_amd.patch_reference()          # before or after `import macarons...`; idempotent
    from macarons.testers.shapenet import run_test      # now builds the HIP SconeOcc / SconeVis

What is swapped:
  * modules  macarons.networks.{Attention, SconeVis, SconeOcc}  ->  macarons_amd.networks.{...}
  * in macarons.networks.Macarons: the class `Macarons` keeps upstream's definition (depth model, optimiser and factories are
    upstream's business) but its `compute_visibility_gains` (Macarons.py:138-178, a second copy of the scorer) is routed to the
    HIP scorer, and its module-level `SconeOcc` / `SconeVis` / block names are the HIP ones, so `create_macarons_model` builds them;
  * helpers (helpers=True): the step either side of the networks in macarons.utility.scone_utils (compute_view_state,
    move_view_state_to_view_space, compute_view_harmonics, compute_occupancy_probability, filter_proxy_points,
    sample_proxy_points), utils.get_knn_points, and in macarons.utility.macarons_utils the per-call occupancy chunker and the three
    distance factors -- same names, same positional arguments, same defaults (tests/test_patch_reference.py checks the signatures);
    helpers="all" adds compute_scene_occupancy_probability_field (one batched pass over the grid cells instead of a Python loop).

Everything else (trainers, testers, data loading, depth network, weight init walkers, checkpoint loaders) is upstream's code running
unchanged over these classes: the weight-init walk (scone_utils.py:399-428) finds the same `nn.Linear` children by the same names, and
`load_ddp_state_dict` (utils.py:140-158) loads reference checkpoints because the state-dict keys and shapes are the reference's.
"""
import importlib
import inspect
import sys
import types

_NETWORK_MODULES = ("Attention", "SconeVis", "SconeOcc")

# reference module (relative to the package) -> (macarons_amd module, names); the "all" tier is opt-in
_HELPERS = {
    "utility.scone_utils": ("macarons_amd.utility.scone_utils",
                            ("compute_view_state", "move_view_state_to_view_space", "compute_view_harmonics",
                             "compute_occupancy_probability", "filter_proxy_points", "sample_proxy_points")),
    "utility.utils": ("macarons_amd.utility.utils", ("get_knn_points",)),
    "utility.macarons_utils": ("macarons_amd.utility.macarons_utils",
                               ("compute_occupancy_probability", "get_distance_factor", "get_distance_factor_threshold",
                                "get_distance_factor_smooth")),
}
_HELPERS_ALL = {
    "utility.macarons_utils": ("macarons_amd.utility.macarons_utils", ("compute_scene_occupancy_probability_field",)),
}

_STATE = {}          # package -> report of the last patch (idempotence, `unpatch_reference`)


def _defined_in(obj, module):
    return (inspect.isclass(obj) or inspect.isfunction(obj)) and getattr(obj, "__module__", None) == module.__name__


def _signature_extends(ref_fn, new_fn):
    """True when new_fn can be called exactly as ref_fn is: same leading parameters (names, order, defaults); new_fn may add
    optional trailing ones, and may give a default where the reference has none."""
    rp, np_ = list(inspect.signature(ref_fn).parameters.values()), list(inspect.signature(new_fn).parameters.values())
    if len(np_) < len(rp):
        return False
    for r, n in zip(rp, np_):
        if r.name != n.name or r.kind != n.kind:
            return False
        if r.default is not inspect.Parameter.empty and r.default != n.default:
            return False
    return all(p.default is not inspect.Parameter.empty for p in np_[len(rp):])


def patch_reference(package="macarons", helpers=True, import_consumers=True):
    """Install the MI355X networks under `<package>.networks.*` and rebind the names the reference already imported.

    package           top-level name of the reference package (importable: on sys.path or installed)
    helpers           True: also swap the helper functions listed in the module docstring; "all": + the batched occupancy field;
                      False: networks only
    import_consumers  import `<package>.utility.scone_utils` and `<package>.networks.Macarons` now (they need PyTorch3D like the
                      rest of the reference) so that their names are bound to the HIP classes when this returns; False leaves that
                      to the caller's own imports (which resolve against the installed modules anyway)
    Returns a report: {"modules": [...], "rebound": [(module, name), ...], "helpers": [(module, name), ...]}.
    Raises ImportError if the reference package cannot be imported, TypeError if a helper's signature drifted from the reference's.
    """
    importlib.import_module(package)                      # ImportError here = the reference is not importable: nothing to patch
    networks_pkg = importlib.import_module(package + ".networks")
    report = {"package": package, "modules": [], "rebound": [], "helpers": [], "originals": {}}
    swap = {}                                              # id(reference object) -> (replacement, keep-alive of the original)

    def plan(orig_mod, new_mod, names=None):
        for name in (names if names is not None else [n for n in vars(new_mod) if not n.startswith("_")]):
            new = getattr(new_mod, name, None)
            old = getattr(orig_mod, name, None)
            if new is None or old is None or old is new or not _defined_in(old, orig_mod):
                continue
            swap[id(old)] = (new, old)

    # 1. the three network modules: MODULE objects (macarons_amd.networks re-exports the classes under the same names, so
    #    `macarons_amd.networks.SconeVis` as an attribute is the CLASS -- importlib returns the module)
    for short in _NETWORK_MODULES:
        new_mod = importlib.import_module("macarons_amd.networks." + short)
        full = f"{package}.networks.{short}"
        orig = sys.modules.get(full)
        if orig is not None and orig is not new_mod:
            report["originals"][full] = orig
            plan(orig, new_mod)
        sys.modules[full] = new_mod
        setattr(networks_pkg, short, new_mod)
        report["modules"].append(full)

    # 2. consumers named by the verdict of the import graph (they bind names at import time)
    consumers = []
    if import_consumers:
        for rel in ("utility.scone_utils", "networks.Macarons"):
            consumers.append(importlib.import_module(f"{package}.{rel}"))

    # 3. helpers
    tiers = [] if not helpers else ([_HELPERS, _HELPERS_ALL] if helpers == "all" else [_HELPERS])
    for tier in tiers:
        for rel, (new_name, names) in tier.items():
            full = f"{package}.{rel}"
            ref_mod = sys.modules.get(full)
            if ref_mod is None:
                if not import_consumers:
                    continue
                ref_mod = importlib.import_module(full)
            new_mod = importlib.import_module(new_name)
            for name in names:
                old, new = getattr(ref_mod, name, None), getattr(new_mod, name)
                if old is None or old is new:
                    continue
                if inspect.isfunction(old) and old.__module__ == ref_mod.__name__:
                    if not _signature_extends(old, new):
                        raise TypeError(f"{new_name}.{name}{inspect.signature(new)} cannot be called like "
                                        f"{full}.{name}{inspect.signature(old)}")
                    swap[id(old)] = (new, old)
                    report["originals"][f"{full}.{name}"] = old
                setattr(ref_mod, name, new)
                report["helpers"].append((full, name))

    # 4. sweep: every module of the reference package that already holds one of the replaced objects gets the replacement
    ours = {id(importlib.import_module("macarons_amd.networks." + s)) for s in _NETWORK_MODULES}
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or id(mod) in ours or not isinstance(mod, types.ModuleType):
            continue
        if mod_name != package and not mod_name.startswith(package + "."):
            continue
        for name, value in list(vars(mod).items()):
            hit = swap.get(id(value))
            if hit is not None and hit[1] is value:
                setattr(mod, name, hit[0])
                report["rebound"].append((mod_name, name))

    # 5. upstream's Macarons wrapper: its own copy of the per-point scorer goes to the HIP one
    mac_mod = sys.modules.get(f"{package}.networks.Macarons")
    if mac_mod is not None and hasattr(mac_mod, "Macarons"):
        from .networks.Macarons import Macarons as _HipMacarons
        ref_cls = mac_mod.Macarons
        if ref_cls is not _HipMacarons and "compute_visibility_gains" in vars(ref_cls):
            if "_mcr_original_compute_visibility_gains" not in vars(ref_cls):
                ref_cls._mcr_original_compute_visibility_gains = vars(ref_cls)["compute_visibility_gains"]
            ref_cls.compute_visibility_gains = _HipMacarons.compute_visibility_gains
            report["rebound"].append((mac_mod.__name__, "Macarons.compute_visibility_gains"))

    prev = _STATE.get(package)
    if prev is not None:                                   # keep the FIRST originals: a second call sees our objects as "original"
        merged = dict(report["originals"])
        merged.update(prev["originals"])
        report["originals"] = merged
    _STATE[package] = report
    return report


def unpatch_reference(package="macarons"):
    """Undo `patch_reference` as far as module-level names go (tests): reference modules that were imported before the patch are
    re-installed, helpers restored.  Reference modules first imported AFTER the patch hold the HIP classes and keep them."""
    rep = _STATE.pop(package, None)
    if rep is None:
        return
    for short in _NETWORK_MODULES:
        full = f"{package}.networks.{short}"
        orig = rep["originals"].get(full)
        if orig is not None:
            sys.modules[full] = orig
            setattr(sys.modules[package + ".networks"], short, orig)
        else:
            sys.modules.pop(full, None)
    for full, name in rep["helpers"]:
        old = rep["originals"].get(f"{full}.{name}")
        if old is not None and full in sys.modules:
            setattr(sys.modules[full], name, old)
    mac_mod = sys.modules.get(f"{package}.networks.Macarons")
    if mac_mod is not None and hasattr(mac_mod, "Macarons"):
        orig = vars(mac_mod.Macarons).get("_mcr_original_compute_visibility_gains")
        if orig is not None:
            mac_mod.Macarons.compute_visibility_gains = orig
            del mac_mod.Macarons._mcr_original_compute_visibility_gains
