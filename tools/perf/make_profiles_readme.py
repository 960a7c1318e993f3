"""Regenerates profiles/README.md from the committed rocprofv3 / bench artifacts (run from the repo root, no GPU needed)."""
import csv, json, os
P = "profiles"
R = "r02"          # the round whose artifacts this README describes (earlier rounds' files stay in profiles/ as history)
bench = json.loads(open(os.path.join(P, R + "_bench.json")).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(os.path.join(P, R + "_train_step_kernel_stats.csv"))))
pmc = json.load(open(os.path.join(P, R + "_pmc_summary.json")))
other = {}
if os.path.exists(os.path.join(P, R + "_other_configs.jsonl")):
    for line in open(os.path.join(P, R + "_other_configs.jsonl")):
        line = line.strip()
        if line.startswith("{"):
            d = json.loads(line); other[d["config"]] = d
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = []
w = out.append
w("# profiles/ — rocprofv3 evidence, round 2\n")
w("Regenerate this file with `python tools/perf/make_profiles_readme.py`; collect the inputs on the GPU box with `bash tools/perf/pmc_bench.sh`\n"
  "(kernel stats + one PMC counter group per pass) and `python bench.py > profiles/r02_bench.json`.\n")
w("| file | what |\n|---|---|\n"
  "| `r02_bench.json` | the bench line of the un-profiled run (1x MI355X, default flags), extras included (`other_configs`, `small_batch`) |\n"
  "| `r02_train_step_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras` |\n"
  "| `r02_train_step_kernel_stats_no_overlap.csv` | the same with `--no-overlap` (side stream off): the per-kernel averages that bench.py's single-stream live measurement is comparable with |\n"
  "| `r02_pmc_summary.json` | per-kernel averages of the PMC passes over ONE train_step (`FETCH_SIZE`; `WRITE_SIZE`; `SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`) |\n"
  "| `r02_rowln_ablation.txt` | the full-row GEMM+LayerNorm kernel on the two estimator shapes with parts of it switched off (`tools/perf/rowln_bench.py`, `ROWLN_DBG`): where the distance to the MFMA peak goes |\n"
  "| `r02_rowln_pmc.txt` | PMC passes of the same kernel alone (`tools/perf/pmc_rowln.sh`) |\n"
  "| `r02_gemm_mid.txt` | the reverse loop's one-workgroup-per-CU GEMM per shape and split (`tools/perf/mid_bench.py`) |\n"
  "| `r01_*`, `gemm_big_r01.txt`, `gemm_shapes_r01.txt` | round 1's artifacts, kept as history (the kernels they describe still serve cfg 3 / cfg 4 and the BPTT) |\n")
w("## Bench line (cfg 2: B=256, H=5, K=10, S=10, N=50, D=512, fp32)\n")
w(f"* **{bench['value']:.2f} train steps/s** = {bench['ms_per_step']:.1f} ms/step; whole-step algorithmic rate {bench['train_step_algorithmic_tflops']:.1f} TFLOP/s "
  f"= {100*bench['train_step_frac_of_f32_mfma_peak']:.1f} % of the 157.3 TFLOP/s fp32-MFMA peak; reverse loop alone {bench['denoise_steps_per_s']:.0f} steps/s at B=256.")
for key in ("roofline", "roofline_second_kernel"):
    r = bench[key]
    w(f"* `{key}`: `{r['kernel']}` — {r['achieved']:.1f} TFLOP/s = **{100*r['frac']:.1f} %** of peak over {r['launches_per_step']} launches "
      f"({r['ms_per_step']:.1f} ms of the profiled step, {r['us_per_launch']:.1f} µs and {r['flop_per_launch']/1e9:.2f} GFLOP per launch"
      + (f", PMC traffic {r['traffic']/1e6:.1f} MB per launch" if r.get("traffic") else "") + ").")
r = bench["roofline"]
if r.get("traffic") and r.get("algorithmic_bytes_per_launch"):
    w(f"  PMC traffic vs algorithmic bytes of the dominant kernel (`roofline`): {r['traffic']/1e6:.0f} MB against {r['algorithmic_bytes_per_launch']/1e6:.0f} MB per launch "
      f"({r['traffic']/r['algorithmic_bytes_per_launch']:.2f}x; algorithmic = A + W + skip operand read, output written, 4 B each).  `FETCH_SIZE` counts what the "
      "eight per-XCD L2s pull from the fabric (Infinity Cache or HBM): the 4 MB weight matrix is fetched by every XCD and falls out of its 4 MB L2 now and then "
      f"(L2 hit rate 93 %, `r02_rowln_pmc.txt`); the write side equals the algorithmic figure exactly.  At {r['traffic']/r['us_per_launch']/1e6:.2f} TB/s the kernel is far "
      "from the 8 TB/s roof: it is bound by MFMA issue (pipe 79 % busy at the 2.08 GHz the chip holds under this load, `r02_pmc_summary.json`).")
for key in ("roofline_big_family", "roofline_small_family"):
    r = bench[key]
    w(f"* `{key}`: {r['kernel']} — {r['achieved']:.1f} TFLOP/s = {100*r['frac']:.1f} % over {r['launches_per_step']} launches, {r['ms_per_step']:.1f} ms.")
f = bench["roofline_gemm_family"]; h = bench["roofline_hbm"]
w(f"* every GEMM launch: {f['achieved']:.1f} TFLOP/s = {100*f['frac']:.1f} % ({f['launches_per_step']} launches, {f['flop_per_step']/1e12:.2f} TFLOP per step).")
w(f"* HBM-bound kernel (`adamw_kernel`, 1.09 GB of algorithmic traffic per launch over the 27.15 M-parameter score arena): {h['achieved']:.0f} GB/s = {100*h['frac']:.0f} % of the 8 TB/s peak "
  f"(≈{100*h['achieved']/6300:.0f} % of the 6.3 TB/s the guide measures as achievable).")
if "cpu_baseline" in bench:
    c = bench["cpu_baseline"]
    w(f"* CPU baseline (oracle port, {c['cores']} host cores): {c['value']:.3f} steps/s ({c['sample']}) → GPU/CPU = {bench['speedup_vs_cpu_baseline']:.0f}×.")
w("")
noov = os.path.join(P, R + "_train_step_kernel_stats_no_overlap.csv")
if os.path.exists(noov):
    nrows = {r["Name"]: r for r in csv.DictReader(open(noov))}
    w("## Live HIP-event measurement (bench.py, one single-stream step) vs rocprofv3 (`bench.py --no-overlap`, whole process)\n")
    w("| kernel instantiation | launches/step | bench µs/launch | bench TFLOP/s | rocprofv3 avg µs | rocprofv3 calls |\n|---|---|---|---|---|---|")
    for r in bench.get("roofline_kernels", []):
        short = r["kernel"].replace("aid::", "")
        m = [v for k, v in nrows.items() if ("aid::" + short + "(") in k]
        ra = f"{float(m[0]['AverageNs'])/1e3:.1f}" if m else "—"; rc = m[0]["Calls"] if m else "—"
        w(f"| `{short}` | {r['launches_per_step']} | {r['us_per_launch']:.1f} | {r['achieved']:.1f} | {ra} | {rc} |")
    w("\nThe rocprofv3 column averages over every call of the process (warm-up, timed steps, the denoise-only loop, the profiled step), so the shape\n"
      "mix differs slightly from one train_step; the two columns agree to within that mix.\n")
w(f"## Kernel time under rocprofv3 (sum of kernel durations {tot/1e6:.0f} ms over the whole bench process; two streams overlap, so the sum exceeds wall time)\n")
w("| kernel | calls | avg µs | total ms | % |\n|---|---|---|---|---|")
for r in rows[:16]:
    w(f"| `{r['Name'][:88]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.1f} | {float(r['Percentage']):.1f} |")
w("")
w("## PMC passes (one train_step; FETCH_SIZE x 1 KiB x 2 = corrected read bytes, WRITE_SIZE x 1 KiB; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs ÷ GRBM_GUI_ACTIVE / 8 XCDs)\n")
w("| kernel | launches | avg µs | read MB/launch | write MB/launch | MFMA pipe busy | eff. clock GHz |\n|---|---|---|---|---|---|---|")
for k, v in pmc.items():
    w(f"| `{k[:60]}` | {v.get('launches_mfma','')} | {v.get('avg_us_mfma',0):.1f} | {v.get('fetch_bytes_per_launch_corrected',0)/1e6:.1f} | {v.get('write_bytes_per_launch',0)/1e6:.1f} | "
      f"{100*v.get('mfma_util',0):.1f} % | {v.get('eff_clock_GHz',0):.2f} |")
w("\nThe effective-clock column reads high on dispatches shorter than ≈0.3 ms (guide, DVFS note); it is meaningful for the millisecond-long `gemm_rowln_kernel` rows only.\n")
oc = bench.get("other_configs", {})
if oc:
    w("## Other BASELINE configs at full size (same build, same process as the bench line, outside its timed region)\n")
    for k, d in oc.items():
        w(f"* {k}: {d['ms_per_step']:.1f} ms/step = {d['steps_per_s']:.2f} steps/s, {d['algorithmic_tflops']:.1f} algorithmic TFLOP/s = {100*d['frac_of_f32_mfma_peak']:.1f} % of the fp32-MFMA peak")
    w("")
sb = bench.get("small_batch", {})
if sb:
    w("## Small-batch reverse loops (hipGraph replay; bound by streaming the score network's weights once per denoise step)\n")
    for k, d in sb.items():
        w(f"* {k}: {d['rows']} rows, {d['us_per_denoise_step']:.1f} µs per denoise step = {d['denoise_steps_per_s']:.0f} steps/s; "
          f"{d['weight_bytes_per_step']/1e6:.1f} MB of weights per step -> {d['achieved_GBps']:.0f} GB/s = {100*d['achieved_GBps']/d['peak_GBps']:.1f} % of 8 TB/s")
    w("")
open(os.path.join(P, "README.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out)[:3000])
