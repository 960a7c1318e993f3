#!/bin/bash
# PMC passes over the big-GEMM microbenchmark (one counter group per pass, as the guide prescribes). Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gemm; mkdir -p $OUT
SH="${1:-25600x1024x1024}"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/p1 --output-format csv -- python3 tools/perf/big_gemm.py $SH > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/p2 --output-format csv -- python3 tools/perf/big_gemm.py $SH > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p3 --output-format csv -- python3 tools/perf/big_gemm.py $SH > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p4 --output-format csv -- python3 tools/perf/big_gemm.py $SH > $OUT/p4.log 2>&1
find $OUT -name "*.csv" | head -20
