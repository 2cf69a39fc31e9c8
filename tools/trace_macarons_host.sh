#!/bin/bash
# Dev: host timeline of one MACARONS decision from a HIP API trace: where the host thread sits between / inside API calls.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mh; timeout -s KILL 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/mh -o t -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import torch, bench
print(bench.measure_macarons_step(torch.device('cuda:0'))['p50_ms'])" > /tmp/mh.log 2>&1
tail -1 /tmp/mh.log
python - <<'PY'
import csv, glob, collections
api = glob.glob("/tmp/mh/**/t_hip_api_trace.csv", recursive=True)[0]
ker = glob.glob("/tmp/mh/**/t_kernel_trace.csv", recursive=True)[0]
K = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Correlation_Id"])) for r in csv.DictReader(open(ker))))
A = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], int(r["Correlation_Id"]), int(r["Thread_Id"])) for r in csv.DictReader(open(api))))
main_tid = collections.Counter(a[4] for a in A).most_common(1)[0][0]
A = [a for a in A if a[4] == main_tid]
kname = {k[3]: k[2] for k in K}
# decisions: launches of proxy_update_kernel
marks = [a[0] for a in A if "proxy_update_kernel" in kname.get(a[3], "")]
spans = sorted((b - a, a, b) for a, b in zip(marks[3:-1], marks[4:]))
d, t0, t1 = spans[len(spans) // 2]
seg = [a for a in A if t0 <= a[0] < t1]
print(f"median decision (host, launch to launch): {d/1e6:.2f} ms, {len(seg)} HIP API calls")
inside = collections.defaultdict(lambda: [0, 0])
for s, e, f, c, _ in seg: inside[f][0] += e - s; inside[f][1] += 1
print(" time INSIDE API calls:")
for f, (t, n) in sorted(inside.items(), key=lambda kv: -kv[1][0])[:8]: print(f"   {t/1e6:8.3f} ms x{n:5d}  {f}")
tot_in = sum(v[0] for v in inside.values())
print(f" total inside {tot_in/1e6:.2f} ms; between calls (Python / torch dispatch) {(d - tot_in)/1e6:.2f} ms")
gaps = []
for p, n in zip(seg[:-1], seg[1:]):
    g = n[0] - p[1]
    if g > 0: gaps.append((g, kname.get(p[3], p[2])[:46], kname.get(n[3], n[2])[:46]))
gaps.sort(reverse=True)
print(" largest host gaps between API calls:")
for g, a, b in gaps[:16]: print(f"   {g/1e3:8.1f} us  after {a:46s} before {b}")
print(f" sum of gaps > 100 us: {sum(g for g, _, _ in gaps if g > 1e5)/1e6:.2f} ms; gaps 20-100 us: {sum(g for g, _, _ in gaps if 2e4 < g <= 1e5)/1e6:.2f} ms; < 20 us: {sum(g for g, _, _ in gaps if g <= 2e4)/1e6:.2f} ms")
PY
