#!/bin/bash
# PMC passes for the fused local transformer (VARIANT env); summaries land in gpurun_out/pmc_lp${VARIANT}/
V=${VARIANT:-3}
OUT=/root/repo/gpurun_out/pmc_lp$V
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT -o a -- python /root/repo/tools/time_local_pct.py > $OUT.a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o b -- python /root/repo/tools/time_local_pct.py > $OUT.b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LEVEL_WAVES --output-format csv -d $OUT -o c -- python /root/repo/tools/time_local_pct.py > $OUT.c.log 2>&1
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    d=collections.defaultdict(list); meta=None
    for r in csv.DictReader(open(f)):
        if 'local_pct' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value'])); meta=(r['LDS_Block_Size'],r['Scratch_Size'],r['VGPR_Count'],r['Accum_VGPR_Count'])
    print(f.split('/')[-1], meta)
    for k,v in d.items(): print("  %-28s %16.0f  per-wave(16384) %10.1f" % (k, sum(v)/len(v), sum(v)/len(v)/16384))
PY
