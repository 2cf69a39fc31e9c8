"""Dev: per-kernel time inside the median NBV step of a rocprofv3 kernel trace (csv); steps = spans between the decisions' read-backs (tools/_trace_steps.py)."""
import csv, sys, collections
raw = list(csv.DictReader(open(sys.argv[1])))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in raw))
def _wg(r):
    t = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, t), t
grid = {int(r["Start_Timestamp"]): _wg(r) for r in raw}
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _trace_steps import step_starts
starts = step_starts(rows)                       # a step = from the first kernel after a decision's read-back to the next read-back
steps = []
for a, b in zip(starts[8:-1], starts[9:]):
    seg = rows[a:b]
    steps.append((max(e for _, e, _ in seg) - seg[0][0], seg))
steps.sort(key=lambda x: x[0])
span, seg = steps[len(steps) // 2]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    k = n.split("(")[0][-60:]
    agg[k][0] += e - s; agg[k][1] += 1
print(f"median step span {span/1e6:.3f} ms, {len(seg)} kernels, sum of kernel times {sum(v[0] for v in agg.values())/1e6:.3f} ms")
for k, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:28]:
    print(f"  {t/1e3:9.1f} us  x{c:3d}  {k}")
if len(sys.argv) > 2:                  # any second argument: the launches that take > 10 us with fewer than 1024 workgroups
    print("---- > 10 us with < 1024 workgroups (workgroups x threads, us, kernel)")
    for s_, e_, n_ in seg:
        g_ = grid.get(s_, (0, 0))
        if e_ - s_ > 10000 and g_[0] < 1024: print(f"  {g_[0]:6d} x {g_[1]:4d}  {(e_-s_)/1e3:8.1f}  {n_[:100]}")
