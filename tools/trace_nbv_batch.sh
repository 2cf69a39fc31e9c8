#!/bin/bash
# Dev: GPU busy / idle inside one scene-batch decision (BASELINE config 3, bench.measure_nbv_batch): kernel trace, steps delimited by
# view_state_kernel; per-kernel totals of the median step
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/btrace; timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/btrace -o t -- python /root/repo/tools/time_nbv_batch.py 16 > /tmp/btrace.log 2>&1 < /dev/null
tail -1 /tmp/btrace.log
T=$(find /tmp/btrace -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/trace_gaps.py $T
python /root/repo/tools/step_breakdown.py $T | head -40
