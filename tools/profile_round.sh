#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats of the same command, per-kernel time inside the median NBV step,
# PMC passes (own runs, --pmc only) for the scorer, the fused local transformer, the head GEMM and the kNN, power telemetry.
# Outputs under gpurun_out/round/ (copy the summaries into profiles/ with the round's prefix).
R=/root/repo
OUT=$R/gpurun_out/round
mkdir -p $OUT
cd $R && python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
cd $R && python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_driver_cmd.err | tail -1 > $OUT/bench_driver_cmd.json   # the driver's own command line
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o bench -- python $R/bench.py --steps 500 --warmup 100 --no-cpu-baseline --nbv-iters 20 > $OUT/kstats.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_scorer -o bench -- python $R/bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-nbv --no-strong --streams 1 > $OUT/kstats_scorer.log 2>&1
rm -rf $OUT/ktrace; timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace -o t -- python $R/tools/run_nbv_steps.py 40 > $OUT/ktrace.log 2>&1
T=$(find $OUT/ktrace -name "*kernel_trace.csv" | head -1)
python $R/tools/step_breakdown.py $T > $OUT/nbv_step_breakdown.txt 2>&1
python $R/tools/trace_gaps.py $T > $OUT/nbv_gaps.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  WPS=0 timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_scorer -o p$i -- python $R/tools/time_scorer.py > $OUT/pmc_scorer.p$i.log 2>&1 || echo "scorer pmc pass $i failed"
done
python - <<PY
import csv, collections, glob, json
d = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pmc_scorer/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if 'sh_gain_kernel' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
res = {"kernel": "sh_gain_kernel<true>", "workload": "N=100000 C=200 B=1", "per_dispatch_mean": {k: sum(v) / len(v) for k, v in d.items()},
       "notes": "rocprofv3 --pmc passes (separate runs, no tracing); SQ_* cycle counters in quad-cycles; FETCH_SIZE in KiB as reported "
                "(gfx950: x2 for wide streaming reads per MI355X_MICROARCH.md)"}
if "FETCH_SIZE" in res["per_dispatch_mean"]:
    res["hbm_read_bytes_per_launch_corrected"] = res["per_dispatch_mean"]["FETCH_SIZE"] * 1024 * 2
json.dump(res, open("$OUT/scorer_pmc.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
VARIANT=6 $R/tools/pmc_local_pct.sh > $OUT/local_pct6_pmc.txt 2>&1
VARIANT=6 $R/tools/pmc_local_pct_mem.sh >> $OUT/local_pct6_pmc.txt 2>&1
rm -rf $R/gpurun_out/pmc_generic; KERNEL=linear3p_kernel $R/tools/pmc_generic.sh python $R/tools/run_nbv_steps.py 12 > $OUT/linear3p_pmc.txt 2>&1
rm -rf $R/gpurun_out/pmc_generic; MS=10240 KERNEL="knn_grid_kernel<16, true, 0>" $R/tools/pmc_generic.sh python $R/tools/time_knn.py > $OUT/knn_pmc.txt 2>&1
rm -rf $R/gpurun_out/pmc_generic; MCR_KNN_GRID=0 MS=10240 KERNEL=knn_mfma_kernel $R/tools/pmc_generic.sh python $R/tools/time_knn.py > $OUT/knn_bruteforce_pmc.txt 2>&1
(for c in shell cube; do for g in 1 0; do CLOUD=$c MCR_KNN_GRID=$g python $R/tools/time_knn.py; done; done) > $OUT/knn_times.txt 2>&1
rm -rf $R/gpurun_out/pmc_generic; REPS=3 KERNEL=attention_planes_kernel $R/tools/pmc_generic.sh python $R/tools/time_attention_planes.py > $OUT/attention_planes_pmc.txt 2>&1
cd $R && python tools/power_trace.py > $OUT/power_trace.txt 2>&1
# ---- variant 7 (the opt-in 16-bit matrix path): PMC of local_pct7_kernel, energy per query beside variant 6, the step's breakdown, the
# single-plane head GEMMs
VARIANT=7 $R/tools/pmc_local_pct.sh > $OUT/local_pct7_pmc.txt 2>&1
VARIANT=7 $R/tools/pmc_local_pct_mem.sh >> $OUT/local_pct7_pmc.txt 2>&1
cd $R && python tools/energy_local_pct7.py > $OUT/power_local_pct7.txt 2>&1
cd /tmp
rm -rf $OUT/ktrace7; VARIANT=7 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace7 -o t -- python $R/tools/run_nbv_steps.py 40 > $OUT/ktrace7.log 2>&1
T7=$(find $OUT/ktrace7 -name "*kernel_trace.csv" | head -1)
(grep p50 $OUT/ktrace7.log; python $R/tools/step_breakdown.py $T7) > $OUT/nbv_step_breakdown_variant7.txt 2>&1
python $R/tools/trace_gaps.py $T7 > $OUT/nbv_gaps_variant7.txt 2>&1
rm -rf $OUT/ktrace7 $OUT/ktrace
rm -rf $R/gpurun_out/pmc_generic; VARIANT=7 KERNEL=linear3p_kernel $R/tools/pmc_generic.sh python $R/tools/run_nbv_steps.py 12 > $OUT/linear3p_variant7_pmc.txt 2>&1
cd $R && python -m pytest tests/test_variant7_gpu.py -q -m gpu -s 2>&1 | grep "^\[variant 7\]\|passed\|failed" > $OUT/variant7_test_report.txt
grep -c . $OUT/local_pct6_pmc.txt $OUT/linear3p_pmc.txt $OUT/knn_pmc.txt $OUT/power_trace.txt
