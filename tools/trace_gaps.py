"""Dev: GPU idle time inside one NBV step from a rocprofv3 kernel trace (csv): sum of gaps between consecutive kernels.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --nbv-iters 20
    python tools/trace_gaps.py out/t_kernel_trace.csv"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
# steps = spans between consecutive view_state_kernel launches
starts = [i for i, r in enumerate(rows) if "view_state_kernel" in r[2]]
res = []
for a, b in zip(starts[10:-1], starts[11:]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = sorted(((seg[i + 1][0] - seg[i][1], seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1)), reverse=True)
    res.append((span, busy, gaps[:6], len(seg)))
res.sort(key=lambda x: x[0])
span, busy, gaps, n = res[len(res) // 2]
print(f"median step: {n} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms")
for g, a, b in gaps:
    print(f"  gap {g/1e3:8.1f} us  after {a}  before {b}")
