#!/bin/bash
# usage: KERNEL=<substring> tools/pmc_mem.sh <command...>   -- memory-side counter passes (own runs, --pmc only): HBM fetch, L2 hit / miss, L1 -> L2 requests
OUT=/root/repo/gpurun_out/pmc_mem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $OUT -o m$i -- "$@" > $OUT.m$i.log 2>&1 || echo "pass $i ($set) failed"
done
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$KERNEL" in r['Kernel_Name']:
            d[(r['Counter_Name'], r['Grid_Size'])].append(float(r['Counter_Value']))
    for (k,gs),v in sorted(d.items()): print("  grid %-9s %-30s mean %16.0f  max %16.0f  n %d" % (gs, k, sum(v)/len(v), max(v), len(v)))
PY
