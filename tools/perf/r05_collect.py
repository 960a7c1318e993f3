"""Gather the summaries of tools/perf/r05_profiles.sh into <out>/collect/ under the names they are committed with in profiles/."""
import glob, os, shutil, sys
out = sys.argv[1]
dst = os.path.join(out, "collect"); os.makedirs(dst, exist_ok=True)
def first(sub, pat):
    f = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return f[0] if f else None
names = {"stats": "r05_train_step_kernel_stats.csv", "stats_noov": "r05_train_step_kernel_stats_no_overlap.csv", "cfg3": "r05_cfg3_kernel_stats.csv",
         "cfg3_noov": "r05_cfg3_kernel_stats_no_overlap.csv", "cfg4": "r05_cfg4_kernel_stats.csv", "cfg4_noov": "r05_cfg4_kernel_stats_no_overlap.csv",
         "small_b1": "r05_small_batch_b1_kernel_stats.csv", "small_b32": "r05_small_batch_b32_kernel_stats.csv"}
for sub, name in names.items():
    f = first(sub, "*kernel_stats.csv")
    if f: shutil.copy(f, os.path.join(dst, name))
for d in sorted(glob.glob(os.path.join(out, "phase_*"))):
    if os.path.isdir(d):
        f = first(os.path.basename(d), "*kernel_stats.csv")
        if f: shutil.copy(f, os.path.join(dst, f"r05_{os.path.basename(d)}_kernel_stats.csv"))
for sub, name in (("pmc_chain", "r05_pmc_chain_summary.json"), ("pmc_tall", "r05_pmc_summary.json")):
    p = os.path.join(out, sub, "pmc_summary.json")
    if os.path.exists(p): shutil.copy(p, os.path.join(dst, name))
for lg in ("stats.log", "stats_noov.log"):
    p = os.path.join(out, lg)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines: open(os.path.join(dst, "r05_bench" + ("_no_overlap" if "noov" in lg else "") + "_under_rocprof.json"), "w").write(lines[-1])
for lg in ("cfg3.log", "cfg3_noov.log", "cfg4.log", "cfg4_noov.log"):
    p = os.path.join(out, lg)
    if os.path.exists(p):
        ms = [l.strip() for l in open(p) if "ms per train_step" in l]
        if ms: open(os.path.join(dst, "r05_" + lg.replace(".log", "_ms_under_rocprof.txt")), "w").write("\n".join(ms) + "\n")
print("collected:", sorted(os.listdir(dst)))
