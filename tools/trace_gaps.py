"""Dev: GPU idle time inside one NBV step from a rocprofv3 kernel trace (csv): the holes in the union of the kernel intervals.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --nbv-iters 20
    python tools/trace_gaps.py out/t_kernel_trace.csv"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
# steps = spans between the decisions' read-backs (tools/_trace_steps.py)
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _trace_steps import step_starts
starts = step_starts(rows)                       # a step = from the first kernel after a decision's read-back to the next read-back
res = []
for a, b in zip(starts[10:-1], starts[11:]):
    seg = rows[a:b]
    # kernels of the step may overlap (SconeOcc's global branch runs on a side stream): busy = length of the UNION of the
    # kernel intervals, gaps = the holes of that union
    span = max(e for _, e, _ in seg) - seg[0][0]
    busy, gaps, cur_end, last = 0, [], seg[0][0], seg[0][2]
    for s_, e_, n_ in seg:
        if s_ > cur_end:
            gaps.append((s_ - cur_end, last[:40], n_[:40]))
            busy += e_ - s_
        else:
            busy += max(0, e_ - cur_end)
        if e_ > cur_end:
            cur_end, last = e_, n_
    gaps.sort(reverse=True)
    res.append((span, busy, gaps[:6], len(seg)))
res.sort(key=lambda x: x[0])
span, busy, gaps, n = res[len(res) // 2]
print(f"median step: {n} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms")
for g, a, b in gaps:
    print(f"  gap {g/1e3:8.1f} us  after {a}  before {b}")
