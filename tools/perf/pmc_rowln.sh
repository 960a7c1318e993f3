#!/bin/bash
# Stall breakdown of gemm_rowln_kernel on one shape: SQ wave-state, LDS, L1 (TCP), L2 (TCC) counters, one group per pass.
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
M="${1:-128000}"; N="${2:-1024}"; K="${3:-1024}"
OUT=gpurun_out/pmc_rowln_${M}x${N}x${K}; rm -rf $OUT; mkdir -p $OUT
P="python3 tools/perf/rowln_one.py $M $N $K 6"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -d $OUT/a --output-format csv -- $P > $OUT/a.log 2>&1; echo a
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/b --output-format csv -- $P > $OUT/b.log 2>&1; echo b
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES -d $OUT/c --output-format csv -- $P > $OUT/c.log 2>&1; echo c
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES GRBM_GUI_ACTIVE -d $OUT/d --output-format csv -- $P > $OUT/d.log 2>&1; echo d
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_READ_REQ TCC_TAG_STALL -d $OUT/e --output-format csv -- $P > $OUT/e.log 2>&1; echo e
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f --output-format csv -- $P > $OUT/f.log 2>&1; echo f
python3 - <<PY
import csv, glob, collections
for sub in "abcdef":
    f = glob.glob(f"$OUT/{sub}/**/*counter_collection.csv", recursive=True)
    if not f: print(sub, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void aid::", "")
        if "gemm_rowln" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in agg:
        print(sub, k, "launches", len(n[k]), "avg_us", round(dur[k] / len(n[k]) / 1e3, 1), {c: round(v / len(n[k])) for c, v in agg[k].items()})
PY
