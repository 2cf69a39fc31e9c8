"""Dev: every host synchronisation inside one MACARONS decision, with its Python call site (torch.cuda.set_sync_debug_mode)."""
import os, sys, warnings, collections, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from macarons_amd.utility import macarons_utils as mu
real = mu.macarons_nbv_decision
sites = collections.Counter()
n = [0]
def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message): return
    st = [f for f in traceback.extract_stack() if "/macarons_amd/" in f.filename]
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-3:]))
    sites[key] += 1
def wrapped(*a, **k):
    n[0] += 1
    if n[0] == 6:
        warnings.showwarning = showwarning
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode(1)
    r = real(*a, **k)
    if n[0] == 6:
        torch.cuda.set_sync_debug_mode(0)
    return r
mu.macarons_nbv_decision = wrapped
bench.measure_macarons_step(torch.device("cuda:0"))
for k, v in sites.most_common(): print(f"{v:3d}  {k}")
print("total", sum(sites.values()))
