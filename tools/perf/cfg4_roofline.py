"""Per-kernel roofline table of BASELINE configs[3] (pixels, B = 256) - VERDICT r4 next-step 4.

Inputs (both from `tools/perf/cfg4_profile.py` with AID_NO_OVERLAP=1, i.e. every phase in order on one stream, so a launch's duration is its own):
  * the per-launch CSV of `AID_DUMP=` (aid_prof_*: one HIP-event pair around every GEMM / implicit-GEMM launch, with its shape)       -> MFMA-bound rows
  * the rocprofv3 --kernel-trace --stats summary of the same program (profiles/rNN_cfg4_kernel_stats_no_overlap.csv)                   -> HBM-bound rows
usage: python tools/perf/cfg4_roofline.py <launches.csv> <kernel_stats.csv> [steps profiled] > profiles/rNN_cfg4_roofline.md"""
import csv, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import bench

PEAK_TF, PEAK_GB = 157.3, 8000.0
B, HW = 256, 42 * 42
MB = 1e6
ACT256 = 256 * B * HW * 4          # one [256 ch, B, 42, 42] fp32 activation: 462 MB
# HBM-bound kernels: algorithmic bytes per call at B = 256 (reads + writes of the tensors the kernel exists to move; parameters and per-row scalars left out)
HBM = {
    "aid::bigln_kernel": (3 * ACT256, "LayerNorm(451 584): x read twice (statistics, normalise), y written"),
    "aid::bigln_bwd_kernel": (5 * ACT256, "dy, x read twice (sums, apply), dx written"),
    "aid::bigln_param_grad_kernel": (2 * ACT256, "dy, x read once; dgamma / dbeta [F] accumulated"),
    "aid::attn_apply_kernel": (2 * ACT256, "x (1 + att): CNHW read, NCHW-flattened written"),
    "aid::attn_pool_kernel": (1 * ACT256, "mean_c / max_c over 256 channels per pixel"),
    "aid::attn_bwd_a_kernel": (3 * ACT256, "dout, x read; dx written; per-pixel sums"),
    "aid::attn_bwd_b_kernel": (2 * ACT256, "dx read-modify-write (avg / argmax routing of the pooled gradient)"),
    "void aid::gn_mish_reg_kernel<8>": (2 * ACT256, "GroupNorm + Mish, layer 3: conv output read once (row in registers), activation written"),
    "void aid::gn_mish_bwd_reg_kernel<8>": (3 * ACT256, "layer 3: conv output, dy read; dx written"),
}


def main():
    launches, stats = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    agg = {}
    for r in csv.DictReader(open(launches)):
        if float(r["ms"]) < 0 or int(r["kernel_id"]) not in (0, 3):
            continue
        name = bench.kernel_name(int(r["variant"]), int(r["layout"]))
        M, N, K = int(r["M"]), int(r["N"]), int(r["K"])
        if M * N * K == 0:
            continue
        e = agg.setdefault((name, M, N, K), [0, 0.0]); e[0] += 1; e[1] += float(r["ms"])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    fl_all = sum(2.0 * k[1] * k[2] * k[3] * v[0] for k, v in agg.items())
    print("# cfg 4 (pixels, B = 256, L = 128, N = 50, H = 5): per-kernel roofline\n")
    print("One train_step, every phase in order on ONE stream (`AID_NO_OVERLAP=1 python tools/perf/cfg4_profile.py`), so a launch's duration is its own.  "
          "MFMA rows: shape and duration of every launch from the library's HIP-event pairs (`AID_DUMP`), grouped by kernel instantiation and shape; "
          f"`frac` = 2MNK / time / {PEAK_TF} TFLOP/s (fp32 MFMA peak, MI355X_MICROARCH.md).  Conv rows: M = Cout, N = B Ho Wo, K = 9 Cin "
          "(weight gradient: M = Cout, N = 9 Cin, K = B Ho Wo).\n")
    print("| share of GEMM time | launches | avg us | GFLOP / launch | TFLOP/s | frac | M x N x K | kernel |")
    print("|---|---|---|---|---|---|---|---|")
    for (name, M, N, K), (n, ms) in rows[:28]:
        fl = 2.0 * M * N * K
        tf = fl * n / ms / 1e9
        print(f"| {100 * ms / tot:.1f} % | {n} | {1e3 * ms / n:.1f} | {fl / 1e9:.2f} | {tf:.1f} | {tf / PEAK_TF:.2f} | {M} x {N} x {K} | `{name}` |")
    print(f"\nGEMM / implicit-GEMM launches of the step: {sum(v[0] for v in agg.values())}, {tot:.1f} ms in all, {fl_all / 1e12:.2f} TFLOP -> "
          f"{fl_all / tot / 1e9:.1f} TFLOP/s over their own time.\n")
    print("## HBM-bound kernels (rocprofv3 --kernel-trace --stats of the same program)\n")
    print(f"`frac` = algorithmic bytes / average duration / {PEAK_GB / 1000:.0f} TB/s (HBM3E peak; ~6.3 TB/s is what a float4 copy achieves on this chip).\n")
    print("| kernel | calls / step | avg us | MB / call | GB/s | frac | what moves |")
    print("|---|---|---|---|---|---|---|")
    st = list(csv.DictReader(open(stats)))
    for r in st:
        nm = r["Name"].split("(")[0].strip()
        if nm in HBM:
            by, what = HBM[nm]
            us = float(r["AverageNs"]) / 1e3
            print(f"| `{nm}` | {int(r['Calls']) / steps:.0f} | {us:.1f} | {by / MB:.0f} | {by / us / 1e3:.0f} | {by / us / 1e3 / PEAK_GB:.2f} | {what} |")
    tot_ns = sum(float(r["TotalDurationNs"]) for r in st)
    print(f"\nAll kernels of the profiled steps: {tot_ns / 1e6 / steps:.1f} ms per step on one stream.")


if __name__ == "__main__":
    main()
