#!/bin/bash
# rocprofv3 passes over bench.py (GPU box, from the repo root): kernel stats, then one PMC group per pass (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r02; rm -rf $OUT; mkdir -p $OUT
# counter collection dies on this pool after a few thousand dispatches in one process (SIGSEGV inside the profiler), so the PMC passes
# profile exactly ONE train_step (3 800 dispatches; same kernels, shapes and traffic as a bench step), single stream
B="tools/perf/pmc_probe2.py step_noov"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
echo stats done
# the same command with the side stream off: per-kernel averages comparable with bench.py's single-stream live measurement (roofline.us_per_launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_noov -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-overlap > $OUT/stats_noov.log 2>&1
echo stats_noov done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $B > $OUT/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $B > $OUT/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $B > $OUT/mfma.log 2>&1
echo mfma done
python3 tools/perf/pmc_reduce.py $OUT
