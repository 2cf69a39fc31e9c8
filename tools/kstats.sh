#!/bin/bash
# Per-kernel time of a command on the GPU box:  tools/kstats.sh NAME [N_ROWS] -- cmd ...   -> gpurun_out/kstats_NAME.txt (also printed)
# (rocprofv3 --kernel-trace --stats, csv output; never reads stdin, never waits on a missing file)
NAME=$1; shift
ROWS=30
if [ "$1" != "--" ]; then ROWS=$1; shift; fi
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kstats_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL ${KSTATS_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- "$@" > $OUT/run.log 2>&1 < /dev/null
python3 - "$OUT" "$ROWS" <<'PY' | tee $R/gpurun_out/kstats_$NAME.txt
import csv, glob, sys
out, rows = sys.argv[1], int(sys.argv[2])
fs = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv under", out); print(open(out + "/run.log").read()[-1500:]); sys.exit(0)
rs = list(csv.DictReader(open(fs[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rs)
print(f"{'calls':>7} {'total us':>10} {'avg us':>9} {'%':>6}  kernel")
for r in rs[:rows]:
    print(f"{int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e3:10.1f} {float(r['AverageNs'])/1e3:9.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}  {r['Name'][:110]}")
PY
