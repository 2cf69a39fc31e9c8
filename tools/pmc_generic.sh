#!/bin/bash
# usage: KERNEL=<substring> tools/pmc_generic.sh <command...>   -- SQ counter passes (own runs, --pmc only)
OUT=/root/repo/gpurun_out/pmc_generic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $OUT -o p$i -- "$@" > $OUT.p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    d=collections.defaultdict(list); g=1
    for r in csv.DictReader(open(f)):
        if "$KERNEL" in r['Kernel_Name']:
            d[(r['Counter_Name'], r['Grid_Size'])].append(float(r['Counter_Value']))
    for (k,gs),v in sorted(d.items()): print("  grid %-9s %-26s %16.0f  per-wave %10.1f" % (gs, k, sum(v)/len(v), sum(v)/len(v)/(int(gs)//64)))
PY
