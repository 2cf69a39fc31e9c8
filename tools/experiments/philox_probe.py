import sys, ctypes, torch
sys.path.insert(0, '/root/repo')
from macarons_amd import ops
dev = torch.device('cuda:0')
for mapping in (1, 0):
    ops._PHILOX_MAPPING = mapping
    torch.manual_seed(1234)
    want = torch.cat([torch.rand(2048, 1, device=dev) for _ in range(3)], 1).t().contiguous()
    off_after = torch.cuda.default_generators[0].get_offset()
    torch.manual_seed(1234)
    got = ops.uniform_rows(3, 2048, dev)
    print("mapping", mapping, "equal", torch.equal(got, want), "maxdiff", float((got - want).abs().max()), "offset after 3 calls", off_after,
          "mine", torch.cuda.default_generators[0].get_offset(), got[0, :3].tolist(), want[0, :3].tolist())
