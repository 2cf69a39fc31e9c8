#!/bin/bash
# SQ counters of one ops.linear shape (M, N, K, GELU env)
OUT=/root/repo/gpurun_out/pmc_lin3
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT -o p$i -- python /root/repo/tools/time_linear_one.py > $OUT.p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    d=collections.defaultdict(list); w=1
    for r in csv.DictReader(open(f)):
        if 'linear' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value'])); g=int(r['Grid_Size'])//64
    for k,v in d.items(): print("  %-28s %16.0f  per-wave(%d) %10.1f" % (k, sum(v)/len(v), g, sum(v)/len(v)/g))
PY
