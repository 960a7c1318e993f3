#!/bin/bash
# rocprofv3 kernel stats of one small-batch reverse loop: sb_prof.sh ROWS [MODE]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_r04
T=sb$1_m${2:--1}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04/$T -- python3 tools/perf/small_batch_prof.py $1 ${2:--1} > gpurun_out/prof_r04/$T.log 2>&1
f=$(find gpurun_out/prof_r04/$T -name "*kernel_stats.csv"); cp $f gpurun_out/prof_r04/${T}_kernel_stats.csv; find gpurun_out/prof_r04/$T -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_r04/${T}_kernel_stats.csv")))
for r in rows[:9]:
    print(f"{int(r['Calls'])/250:6.1f}/step x {float(r['AverageNs'])/1e3:7.2f} us (min {float(r['MinNs'])/1e3:.2f})  {r['Name'][:80]}")
PY
