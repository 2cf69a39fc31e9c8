"""Dev: headline NBV step (Q = 100k, M = 10 240, C = 200), eager sync-free step vs hipGraph replay (nbv.GraphedNbvStep)."""
import sys, os, time, torch, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd.networks import SconeVis, SconeOcc
from macarons_amd.nbv import nbv_step, GraphedNbvStep, ViewStateGrid
dev = torch.device("cuda:0")
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    occ, vis = SconeOcc().to(dev).eval(), SconeVis().to(dev).eval()
with torch.no_grad():
    occ.linear3.bias += 0.5
g = torch.Generator().manual_seed(0)
M, Q, C = 10240, 100000, 200
d = torch.randn(M, 3, generator=g)
pc = (d / d.norm(dim=1, keepdim=True) * torch.tensor([0.35, 0.25, 0.3]) + 0.002 * torch.randn(M, 3, generator=g))[None].to(dev)
X = (torch.rand(1, Q, 3, generator=g) - 0.5).to(dev)
cams = torch.randn(C, 3, generator=g); cams = (1.5 * cams / cams.norm(dim=1, keepdim=True)).to(dev)
xv = cams[:3].contiguous()
grid = ViewStateGrid(dev)
def p50(fn, n=30):
    ts = []
    for _ in range(5): fn()
    torch.cuda.synchronize()
    for _ in range(n):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ts.sort(); return ts[len(ts) // 2] * 1e3
print("eager p50 %.2f ms" % p50(lambda: nbv_step(occ, vis, pc, X, xv, cams, grid)))
gs = GraphedNbvStep(occ, vis, pc, X, xv, cams, grid)
print("graph p50 %.2f ms" % p50(lambda: gs()))
