#!/bin/bash
# Dev: average duration per (kernel, grid) of a command:  tools/ktrace_by_grid.sh <substring filter> -- cmd ...
FILT=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktg; timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktg -o t -- "$@" > /tmp/ktg.log 2>&1 < /dev/null
python3 - "$FILT" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/ktg/**/t_kernel_trace.csv", recursive=True)
if not f: print(open("/tmp/ktg.log").read()[-2000:]); sys.exit(0)
d = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] and sys.argv[1] not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"][:100], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v = sorted(v)
    print(f"{len(v):5d} x  med {v[len(v)//2]/1e3:9.1f} us  grid {k[1]}x{k[2]}x{k[3]} wg {k[4]}  {k[0]}")
PY
