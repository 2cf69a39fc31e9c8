#!/bin/bash
# memory-side PMC passes for the fused local transformer (VARIANT env); <= 4 counters per pass, every pass under timeout
V=${VARIANT:-3}
OUT=/root/repo/gpurun_out/pmc_mem_lp$V
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr" \
           "TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT -o p$i -- python /root/repo/tools/time_local_pct.py > $OUT.p$i.log 2>&1 || echo "pass $i failed: $(grep -m1 -i 'error code' $OUT.p$i.log | cut -c1-200)"
done
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'local_pct' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in d.items(): print("  v$V %-40s %16.0f  per-WG(4096) %12.1f" % (k, sum(v)/len(v), sum(v)/len(v)/4096))
PY
