"""Dev: per-kernel time inside the median NBV step of a rocprofv3 kernel trace (csv); steps = spans between view_state_kernel launches."""
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
starts = [i for i, r in enumerate(rows) if "view_state_kernel" in r[2]]
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
