import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macarons_amd import ops
dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("M", 262144)), int(os.environ.get("N", 128)), int(os.environ.get("K", 128))
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
for _ in range(6): ops.linear(x, w, b, gelu=bool(int(os.environ.get("GELU", 0))))
torch.cuda.synchronize()
