"""Dev: the launch-order timeline (start offset, duration, queue, kernel) of the median NBV step of a rocprofv3 kernel trace (csv);
steps = spans between the decisions' read-backs (tools/_trace_steps.py).   python tools/step_timeline.py out/t_kernel_trace.csv"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _trace_steps import step_starts
starts = step_starts(rows)                       # a step = from the first kernel after a decision's read-back to the next read-back
segs = [(rows[b - 1][1] - rows[a][0], a, b) for a, b in zip(starts[10:-1], starts[11:])]
segs.sort()
span, a, b = segs[len(segs) // 2]
t0 = rows[a][0]
print(f"median segment: {b - a} kernels, {span / 1e3:.1f} us")
for s, e, n, q in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n.split('(')[0][-70:]}")
