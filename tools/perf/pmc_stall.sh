#!/bin/bash
# Stall breakdown of gemm_big_kernel on one MINE shape: SQ wave-state counters, LDS counters, L1 (TCP) and L2 (TCC) counters, one group per pass.
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_stall; rm -rf $OUT; mkdir -p $OUT
SH="${1:-25600x1024x1024}"
export BIG_GEMM_CODES=265
P="python3 tools/perf/big_gemm.py $SH"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -d $OUT/a --output-format csv -- $P > $OUT/a.log 2>&1; echo a
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/b --output-format csv -- $P > $OUT/b.log 2>&1; echo b
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES -d $OUT/c --output-format csv -- $P > $OUT/c.log 2>&1; echo c
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES GRBM_GUI_ACTIVE -d $OUT/d --output-format csv -- $P > $OUT/d.log 2>&1; echo d
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_READ_REQ TCC_TAG_STALL -d $OUT/e --output-format csv -- $P > $OUT/e.log 2>&1; echo e
python3 - <<PY
import csv, glob, collections
for sub in "abcde":
    f = glob.glob(f"$OUT/{sub}/**/*counter_collection.csv", recursive=True)
    if not f: print(sub, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void aid::", "")
        if "gemm_big_kernel<64, 64, 32, false, false, 2>" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        print(sub, k, "launches", len(n[k]), {c: round(v / len(n[k])) for c, v in agg[k].items()})
PY
