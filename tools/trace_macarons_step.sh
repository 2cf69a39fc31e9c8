#!/bin/bash
# Dev: GPU busy / idle inside one MACARONS decision (kernel trace of bench.measure_macarons_step): decisions are delimited by the
# fused depth update (proxy_update_kernel).   [VARIANT=7] tools/trace_macarons_step.sh > gpurun_out/r06_macarons_decision_trace_<tag>.txt
cd /tmp && export TMPDIR=/tmp
export MCR_BENCH_NO_CHECKS=1
rm -rf /tmp/mtrace; timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/mtrace -o t -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import os, contextlib, torch, bench
from macarons_amd import ops
with (ops.variant(int(os.environ['VARIANT'])) if os.environ.get('VARIANT') else contextlib.nullcontext()):     # VARIANT=7: every decision of the trace on the 16-bit path
    r = bench.measure_macarons_step(torch.device('cuda:0'))
print(r['p50_ms'], r['variant_7']['p50_ms'])" > /tmp/mtrace.log 2>&1
tail -1 /tmp/mtrace.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/mtrace/**/t_kernel_trace.csv", recursive=True)[0]
raw = list(csv.DictReader(open(f)))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in raw))
def _wg(r):
    t = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, t), t
grid = {int(r["Start_Timestamp"]): _wg(r) for r in raw}
starts = [i for i, r in enumerate(rows) if "proxy_update_kernel" in r[2]]
res = []
for a, b in zip(starts[3:-1], starts[4:]):
    seg = rows[a:b]
    span = max(e for _, e, _ in seg) - seg[0][0]
    busy, gaps, cur_end, last = 0, [], seg[0][0], seg[0][2]
    for s_, e_, n_ in seg:
        if s_ > cur_end:
            gaps.append((s_ - cur_end, last[:48], n_[:48])); busy += e_ - s_
        else:
            busy += max(0, e_ - cur_end)
        if e_ > cur_end: cur_end, last = e_, n_
    res.append((span, busy, sorted(gaps, reverse=True)[:12], len(seg), seg))
res.sort(key=lambda x: x[0])
span, busy, gaps, n, seg = res[len(res) // 2]
print(f"median decision: {n} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms")
for g, a, b in gaps: print(f"  gap {g/1e3:8.1f} us  after {a}  before {b}")
agg = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in seg: agg[n_[:60]][0] += e_ - s_; agg[n_[:60]][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]: print(f"  {v[0]/1e3:9.1f} us x{v[1]:4d}  {k}")
print("---- launches of the median decision that take > 15 us with fewer than 1024 workgroups (workgroups x threads, us, kernel)")
for s_, e_, n_ in seg:
    g_ = grid.get(s_, (0, 0))
    if e_ - s_ > 15000 and g_[0] < 1024: print(f"  {g_[0]:6d} x {g_[1]:4d}  {(e_-s_)/1e3:8.1f}  {n_[:90]}")
print("---- timeline of the median decision (start us, duration us, gap before us, kernel)")
t0 = seg[0][0]; prev_end = t0
for s_, e_, n_ in seg:
    print(f"{(s_-t0)/1e3:9.1f} {(e_-s_)/1e3:8.1f} {max(0,(s_-prev_end))/1e3:8.1f}  {n_[:70]}")
    prev_end = max(prev_end, e_)
PY
