#!/bin/bash
# Dev: the launch list of ONE SconeVis forward on a single cloud of 2048 tokens (kernel trace of 30 forwards; the last one printed in
# launch order with durations and the gaps in front of each launch).  N_CLOUDS=30: the batch of a decision.
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/_vis1.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from macarons_amd.networks import SconeVis
dev = torch.device("cuda:0"); torch.manual_seed(0)
vis = SconeVis().to(dev).eval()
import os; NC = int(os.environ.get("N_CLOUDS", "1")); pts = torch.rand(NC, 2048, 4, device=dev); vh = torch.randn(NC, 2048, 64, device=dev) * 0.3
with torch.no_grad():
    for _ in range(30):
        vis(pts, view_harmonics=vh); torch.cuda.synchronize()
PY
rm -rf /tmp/kv1; timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kv1 -o t -- python3 /tmp/_vis1.py > /tmp/kv1.log 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kv1/**/t_kernel_trace.csv", recursive=True)
if not f: print(open("/tmp/kv1.log").read()[-2000:]); sys.exit(0)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 30
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"]); prev = t0; busy = 0
print(f"{n} launches per forward")
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += e - s
    print(f"{(s-t0)/1e3:8.1f} us  gap {(s-prev)/1e3:6.1f}  dur {(e-s)/1e3:6.1f}  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {r['Kernel_Name'][:90]}")
    prev = e
print(f"span {(prev-t0)/1e3:.1f} us, busy {busy/1e3:.1f} us")
PY
