"""Gather the summaries of tools/perf/r03_profiles.sh into <out>/collect/ under the names they are committed with in profiles/."""
import glob, os, shutil, sys
out = sys.argv[1]
dst = os.path.join(out, "collect"); os.makedirs(dst, exist_ok=True)
def first(sub, pat):
    f = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return f[0] if f else None
for sub, name in (("stats", "r03_train_step_kernel_stats.csv"), ("stats_noov", "r03_train_step_kernel_stats_no_overlap.csv")):
    f = first(sub, "*kernel_stats.csv")
    if f: shutil.copy(f, os.path.join(dst, name))
for d in sorted(glob.glob(os.path.join(out, "phase_*"))):
    if os.path.isdir(d):
        f = first(os.path.basename(d), "*kernel_stats.csv")
        if f: shutil.copy(f, os.path.join(dst, f"r03_{os.path.basename(d)}_kernel_stats.csv"))
if os.path.exists(os.path.join(out, "pmc_summary.json")):
    shutil.copy(os.path.join(out, "pmc_summary.json"), os.path.join(dst, "r03_pmc_chain_summary.json"))
for lg in ("stats.log", "stats_noov.log"):
    p = os.path.join(out, lg)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines: open(os.path.join(dst, "r03_bench" + ("_no_overlap" if "noov" in lg else "") + "_under_rocprof.json"), "w").write(lines[-1])
print("collected:", sorted(os.listdir(dst)))
