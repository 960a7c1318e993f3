#!/bin/bash
# MFMA-pipe utilisation and effective clock of gemm_rowln_kernel and of its ablation variants (aid_rowln_debug modes), one rocprofv3 pass each.
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
M="${1:-128000}"; N="${2:-1024}"; K="${3:-1024}"
OUT=gpurun_out/pmc_rowln_clock; rm -rf $OUT; mkdir -p $OUT
for mode in 0 16 2 4 3; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/m$mode --output-format csv -- python3 tools/perf/rowln_one.py $M $N $K 6 $mode > $OUT/m$mode.log 2>&1; echo mode $mode
done
python3 - <<PY
import csv, glob, collections
for mode in (0, 16, 2, 4, 3):
    f = glob.glob(f"$OUT/m{mode}/**/*counter_collection.csv", recursive=True)
    if not f: print(mode, "no output"); continue
    agg = collections.defaultdict(float); n = set(); dur = 0.0
    for r in csv.DictReader(open(f[0])):
        if "gemm_rowln" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n:
            n.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    L = len(n); us = dur / L / 1e3
    busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"] / L / 1024; act = agg["GRBM_GUI_ACTIVE"] / L / 8
    print(f"mode {mode:2d}: {L} launches, {us:8.1f} us, {2.0*$M*$N*$K/us/1e6:6.1f} TFLOP/s, MFMA pipe busy {100*busy/act:5.1f} %, clock {act/us/1e3:5.2f} GHz")
PY
