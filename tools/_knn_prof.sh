cd /tmp && export TMPDIR=/tmp
for c in shell cube; do
CLOUD=$c MS=${MS:-10240} timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk$c -o t -- python /root/repo/tools/time_knn.py > /tmp/ok.log 2>&1
f=$(find /tmp/pk$c -name "*kernel_stats.csv" | head -1)
echo "== $c"; python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:9]:
    print("%-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
