#!/bin/bash
# rocprofv3 --kernel-trace --stats of one cfg-3 and one cfg-4 process (3 warm-up + 5 timed train_steps each); GPU box, from the repo root.
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r04; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4 -- python3 tools/perf/cfg4_profile.py 5 > $OUT/cfg4.log 2>&1
echo cfg4 done; grep 'ms per' $OUT/cfg4.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3 -- python3 tools/perf/cfg3_profile.py 5 > $OUT/cfg3.log 2>&1
echo cfg3 done; grep 'ms per' $OUT/cfg3.log
export AID_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4_noov -- python3 tools/perf/cfg4_profile.py 5 > $OUT/cfg4_noov.log 2>&1
echo cfg4_noov done; grep "ms per" $OUT/cfg4_noov.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3_noov -- python3 tools/perf/cfg3_profile.py 5 > $OUT/cfg3_noov.log 2>&1
echo cfg3_noov done; grep "ms per" $OUT/cfg3_noov.log
unset AID_NO_OVERLAP
for c in cfg3 cfg4 cfg3_noov cfg4_noov; do f=$(find $OUT/$c -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/${c}_kernel_stats.csv; done
find $OUT -name '*kernel_trace.csv' -delete
ls -la $OUT
