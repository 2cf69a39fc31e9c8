#!/bin/bash
# Copy the summaries of tools/profile_round.sh / trace_macarons_step.sh (gpurun_out/, scratch) into profiles/ under the round's prefix.
#   tools/copy_round_profiles.sh r05
P=${1:?round prefix}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/round
cp $O/bench.json $R/profiles/${P}_bench_full.json
[ -s $O/bench_driver_cmd.json ] && cp $O/bench_driver_cmd.json $R/profiles/${P}_bench_driver_cmd.json
cp $(find $O/kstats -name "*kernel_stats.csv" | head -1) $R/profiles/${P}_bench_kernel_stats.csv
cp $(find $O/kstats_scorer -name "*kernel_stats.csv" | head -1) $R/profiles/${P}_scorer_only_kernel_stats.csv
cp $O/nbv_step_breakdown.txt $R/profiles/${P}_nbv_step_breakdown.txt
cp $O/nbv_gaps.txt $R/profiles/${P}_nbv_step_gaps.txt
cp $O/scorer_pmc.json $R/profiles/${P}_scorer_pmc.json
cp $O/local_pct6_pmc.txt $R/profiles/${P}_local_pct6_pmc.txt
cp $O/linear3p_pmc.txt $R/profiles/${P}_linear3p_pmc.txt
cp $O/knn_pmc.txt $R/profiles/${P}_knn_pmc.txt
cp $O/knn_bruteforce_pmc.txt $R/profiles/${P}_knn_bruteforce_pmc.txt
cp $O/knn_times.txt $R/profiles/${P}_knn_times.txt
cp $O/power_trace.txt $R/profiles/${P}_power_local_pct6.txt
for k in host device; do
  [ -f $R/gpurun_out/${P}_macarons_decision_trace_$k.txt ] && cp $R/gpurun_out/${P}_macarons_decision_trace_$k.txt $R/profiles/
done
for f in decision_host_timeline scone_vis_batch_launches scone_vis_single_launches; do
  [ -f $R/gpurun_out/${P}_$f.txt ] && cp $R/gpurun_out/${P}_$f.txt $R/profiles/
done
for f in local_pct7_pmc power_local_pct7 nbv_step_breakdown_variant7 linear3p_variant7_pmc variant7_test_report; do
  [ -s $O/$f.txt ] && cp $O/$f.txt $R/profiles/${P}_$f.txt
done
[ -s $O/nbv_gaps_variant7.txt ] && cp $O/nbv_gaps_variant7.txt $R/profiles/${P}_nbv_step_gaps_variant7.txt
[ -f $O/attention_planes_pmc.txt ] && cp $O/attention_planes_pmc.txt $R/profiles/${P}_attention_planes_pmc.txt
ls -la $R/profiles | grep ${P}_ | wc -l
