"""Regenerates profiles/README.md from the committed rocprofv3 / bench artifacts of round 5 (run from the repo root, no GPU needed)."""
import csv, json, os
P = "profiles"
R = "r05"          # the round whose artifacts this README describes (earlier rounds' files stay in profiles/ as history)
bench = json.load(open(os.path.join(P, R + "_bench_detail.json")))          # the full result bench.py writes next to its compact stdout line (R_bench.json)
rows = list(csv.DictReader(open(os.path.join(P, R + "_train_step_kernel_stats.csv"))))
pmc = json.load(open(os.path.join(P, R + "_pmc_chain_summary.json")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = []
w = out.append
short = lambda n: n.replace("void ", "").replace("aid::", "").split("(")[0]
w("# profiles/ — rocprofv3 evidence, round 5\n")
w("Regenerate this file with `python tools/perf/make_profiles_readme.py`; collect the inputs on the GPU box with `bash tools/perf/r05_profiles.sh`\n"
  "(kernel stats of the bench command in both schedules, kernel stats per phase, PMC passes over one 256-row reverse loop) and\n"
  "`python bench.py > profiles/r05_bench.json`.\n")
w("| file | what |\n|---|---|\n"
  "| `r05_bench.json` | the ONE compact stdout line of the un-profiled run (1x MI355X, `python bench.py --steps 20 --warmup 5`: what the driver parses; < 8 KB) |\n"
  "| `r05_bench_detail.json` | `bench_detail.json` of the same run: everything else (`roofline_phases`, `roofline_kernels`, `other_configs` incl. per-phase rooflines of cfg 3 / cfg 4 and the estimator-on pixel step, `small_batch`) |\n"
  "| `r05_cfg4_roofline.md`, `r05_cfg4_launches.csv` | per-kernel roofline table of cfg 4 (pixels, B = 256): every GEMM / implicit-GEMM launch of one single-stream step with its shape and HIP-event duration (`AID_DUMP`), and the HBM-bound kernels from `r05_cfg4_kernel_stats_no_overlap.csv` (`tools/perf/cfg4_roofline.py`) |\n"
  "| `r05_train_step_kernel_stats.csv` | `AID_BENCH_SKIP=phases,denoise,graphs rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras` (round 5: the untimed per-phase / denoise-only measurements are skipped inside the traced process, so a kernel's share is its share in the schedules the line times): both timed regions (side stream + three / two concurrent 256-row chains).  bench.py reads the newest round's file of this name to choose `roofline.kernel` by share of kernel time |\n"
  "| `r05_train_step_kernel_stats_no_overlap.csv` | the same with `--no-overlap` (every phase on one stream): the averages `roofline_single_stream*` is comparable with |\n"
  "| `r05_phase_<phase>_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 tools/perf/phase_profile.py <phase>_cold 5`: ONE phase of the train_step in a cold process (5 calls: the phase's kernels only).  Phases: `elbo`, `rollout_bptt`, `efe` (rollout + estimator + BPTT; estimator = efe - rollout_bptt), `chain1` (one 256-row reverse loop), `chains3` (three concurrent), `chains2` (two: the live belief updates of a step), `value_dynamics`, `optimiser`, `pixel_epistemic` (SURVEY 8(f)-1 at 64 belief rows, S = 2) |\n"
  "| `r05_pmc_chain_summary.json` | per-kernel averages of three PMC passes (`FETCH_SIZE`; `WRITE_SIZE`; `SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`) over ONE 256-row reverse loop alone, launched directly: the kernels of the concurrent chains.  Counter collection on this pool does not survive concurrent streams; one chain alone runs exactly the kernels, shapes and launch geometry each of the three chains runs |\n"
  "| `r05_pmc_summary.json` | the same three passes over ONE policy phase alone (rollout + estimator + BPTT, `phase_profile.py efe_cold 1`): the estimator's tall full-row kernels (replaces `r02_pmc_summary.json`) |\n"
  "| `r05_phase_efe_x6_kernel_stats.csv`, `r05_pmc_x6_summary.json` | the policy phase again with the OPT-IN six-term bf16 arithmetic (`AID_GEMM_X6=1`, `tools/perf/r05_profiles.sh x6`): kernel stats of 5 calls (23.8 ms per call against 29.0 on the float32 pipe) and the MFMA-busy / clock pass (`gemm_x6p_kernel<4, 4, ...>`: 0.54 busy at 1.89 GHz; `gemm_rowln_x6_kernel<1024, ...>`, the form that splits the weights in registers: 0.51 busy at 1.56 GHz - the chip lowers its clock under the bf16 pipe + conversion arithmetic) |\n"
  "| `r05_cfg3_kernel_stats*.csv`, `r05_cfg4_kernel_stats*.csv` (+ `*_ms_under_rocprof.txt`) | `rocprofv3 --kernel-trace --stats -- python3 tools/perf/cfg{3,4}_profile.py 5`: 3 warm-up + 5 timed train_steps of BASELINE configs[2] / [3] at full size, overlapped and single-stream (`_no_overlap`) |\n"
  "| `r05_small_batch_b{1,32}_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 tools/perf/small_batch_prof.py B`: five 50-step reverse loops at B rows, eval mode, direct launches |\n"
  "| `r05_bench_under_rocprof.json`, `r05_bench_no_overlap_under_rocprof.json` | the bench lines printed by the two profiled runs (slower: the profiler serialises the small kernels) |\n"
  "| `r03_*`, `r02_*`, `r01_*`, `gemm_*_r01.txt` | earlier rounds' artifacts, kept as history |\n")
w("## Bench line (cfg 2: B=256, H=5, K=10, S=10, N=50, D=512, fp32)\n")
w(f"* **{bench['value']:.2f} train steps/s** = {bench['ms_per_step']:.1f} ms/step (the step as the reference executes it, dead next_obs belief update included); live-work algorithmic rate "
  f"{bench['train_step_algorithmic_tflops']:.1f} TFLOP/s = {100*bench['train_step_frac_of_f32_mfma_peak']:.1f} % of the 157.3 TFLOP/s fp32-MFMA peak; reverse loop alone {bench['denoise_steps_per_s']:.0f} steps/s at B=256.")
w(f"* `value_dead_belief_elided`: **{bench['value_dead_belief_elided']:.2f} train steps/s** = {bench['ms_per_step_dead_belief_elided']:.1f} ms/step (second timed region of the same run: the agent's default, "
  f"which does not run the belief update `state_agent.py:138-139` computes and `:195-196` overwrites) = {100*bench['train_step_frac_of_f32_mfma_peak_dead_belief_elided']:.1f} % of peak on the same live {bench['train_step_live_flop']/1e12:.2f} TFLOP.")
if bench.get("value_bf16x6"):
    w(f"* opt-in arithmetic (`aid_amd.set_float32_matmul_emulation('bf16x6')`; third and fourth timed regions of the same run): `value_bf16x6` **{bench['value_bf16x6']:.2f} train steps/s** = "
      f"{bench['ms_per_step_bf16x6']:.1f} ms/step as the reference executes the step, `value_bf16x6_dead_belief_elided` **{bench['value_bf16x6_dead_belief_elided']:.2f}** = "
      f"{bench['ms_per_step_bf16x6_dead_belief_elided']:.1f} ms/step in the default schedule: the tall full-row GEMMs as six bf16 MFMA terms of exactly split float32 operands "
      "(float32 accumulation; error against float64 below the float32 MFMA kernels', `tests/test_hip_kernels.py::test_rowln_x6_is_at_least_as_accurate_as_the_float32_mfma`).  Never part of `value`.")
for key in ("roofline", "roofline_second_kernel", "roofline_single_stream"):
    r = bench.get(key)
    if not r:
        continue
    ov = r.get("overlapped_schedule")
    w(f"* `{key}`: `{r['kernel']}` — live (HIP events in this run) {r['achieved']:.1f} TFLOP/s = **{100*r['frac']:.1f} %** of peak, {r['us_per_launch']:.1f} µs and {r['flop_per_launch']/1e9:.2f} GFLOP per launch "
      f"({r['launches_per_step']} launches per step; {r.get('profiled', '')})"
      + (f"; at the committed rocprofv3 average ({r['us_per_launch_rocprof']:.2f} µs, `{r['rocprof_source']}`): {r['achieved_rocprof']:.1f} TFLOP/s = **{100*r['frac_rocprof']:.1f} %** (an event pair adds ~2 µs to a launch of a few µs)" if r.get("frac_rocprof") else "")
      + (f"; in the timed schedule {100*ov['share_of_kernel_time']:.1f} % of kernel time, {ov['calls']} calls x {ov['avg_us']:.2f} µs" if ov else "")
      + (f"; PMC traffic {r['traffic']/1e6:.2f} MB per launch ({r['traffic_source']}) against {r['algorithmic_bytes_per_launch']/1e6:.2f} MB algorithmic" if r.get("traffic") else "") + ".")
f = bench["roofline_gemm_family"]; h = bench["roofline_hbm"]
w(f"* every GEMM launch of one single-stream step: {f['achieved']:.1f} TFLOP/s = {100*f['frac']:.1f} % ({f['launches_per_step']} launches, {f['flop_per_step']/1e12:.2f} TFLOP per step).")
w(f"* HBM-bound kernel (`adamw_kernel` over the 27.15 M-parameter score arena): {h['achieved']:.0f} GB/s = {100*h['frac']:.0f} % of the 8 TB/s peak inside the step.")
if "cpu_baseline" in bench:
    c = bench["cpu_baseline"]
    w(f"* CPU baseline (oracle port, {c['cores']} host cores): {c['value']:.3f} steps/s ({c['sample']}) → GPU/CPU = {bench['speedup_vs_cpu_baseline']:.0f}×.")
w("")
ph = bench.get("roofline_phases") or {}
if ph:
    w("## Per-phase roofline (`roofline_phases`: each phase alone, HIP events on the launching stream, median of 3) next to the rocprofv3 phase summaries\n")
    w("| phase | ms (bench) | work | achieved | % of peak | rocprofv3: kernel ms per call / launches per call (`r05_phase_*`) |\n|---|---|---|---|---|---|")
    names = {"elbo": "elbo", "rollout_bptt": "rollout_bptt", "estimator": None, "reverse_loop_one_chain": "chain1", "reverse_loops_three_chains": "chains3", "reverse_loops_two_chains": "chains2",
             "value_dynamics": "value_dynamics", "optimiser_score": "optimiser"}
    percall = {}
    for k, fn in names.items():
        pth = os.path.join(P, f"{R}_phase_{fn}_kernel_stats.csv") if fn else None
        if pth and os.path.exists(pth):
            rr = [r for r in csv.DictReader(open(pth)) if "aid::" in r["Name"]]
            percall[k] = (sum(float(r["TotalDurationNs"]) for r in rr) / 5e6, sum(int(r["Calls"]) for r in rr) / 5.0)
    pe = os.path.join(P, f"{R}_phase_efe_kernel_stats.csv"); pr = os.path.join(P, f"{R}_phase_rollout_bptt_kernel_stats.csv")
    if os.path.exists(pe) and os.path.exists(pr):
        e = [r for r in csv.DictReader(open(pe)) if "aid::" in r["Name"]]
        percall["estimator"] = (sum(float(r["TotalDurationNs"]) for r in e) / 5e6 - percall["rollout_bptt"][0], sum(int(r["Calls"]) for r in e) / 5.0 - percall["rollout_bptt"][1])
    for k, v in ph.items():
        work = f"{v['flop']/1e9:.0f} GFLOP" if "flop" in v else f"{v['bytes']/1e9:.2f} GB"
        pc = percall.get(k)
        w(f"| `{k}` | {v['ms']:.2f} | {work} | {v['achieved']:.1f} {v['unit']} | **{100*v['frac']:.1f}** | " + (f"{pc[0]:.2f} ms / {pc[1]:.0f}" if pc else "—") + " |")
    w("\nThe rocprofv3 column is the SUM of kernel durations (no gaps between dependent launches; for `chains3` the three streams' kernels overlap in time, so their sum\n"
      "exceeds the phase's wall time).  A chain kernel has the same duration whether one chain runs or three (`gemm_mid_kernel_o3<1,...,0,false>`: "
      + " / ".join(f"{float(r['AverageNs'])/1e3:.2f} µs" for fn in ("chain1", "chains3") for r in csv.DictReader(open(os.path.join(P, f"{R}_phase_{fn}_kernel_stats.csv")))
                   if short(r["Name"]).startswith("gemm_mid_kernel_o3<1, 2, 2, 2, 0, false>"))
      + "): what the concurrency stretches is the gap between a chain's dependent dispatches.  (The chain phases' first call also runs the loop once directly and once\n"
        "under hipGraph capture, so their per-call figures include a fifth of two extra loops.)\n")
w(f"## Kernel time under rocprofv3, timed schedule (sum of kernel durations {tot/1e6:.0f} ms over the whole bench process; streams overlap, so the sum exceeds wall time)\n")
w("| kernel | calls | avg µs | total ms | % |\n|---|---|---|---|---|")
for r in rows[:16]:
    w(f"| `{short(r['Name'])[:88]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.1f} | {float(r['Percentage']):.1f} |")
w("")
noov = os.path.join(P, R + "_train_step_kernel_stats_no_overlap.csv")
if os.path.exists(noov):
    nrows = {short(r["Name"]): r for r in csv.DictReader(open(noov))}
    w("## Live HIP-event measurement (bench.py, one single-stream step) vs rocprofv3 (`bench.py --no-overlap`, whole process)\n")
    w("| kernel instantiation | launches/step | bench µs/launch | bench TFLOP/s | rocprofv3 avg µs | rocprofv3 calls |\n|---|---|---|---|---|---|")
    for r in bench.get("roofline_single_stream_kernels", []):
        k = r["kernel"].replace("aid::", "")
        m = nrows.get(k)
        w(f"| `{k}` | {r['launches_per_step']} | {r['us_per_launch']:.1f} | {r['achieved']:.1f} | " + (f"{float(m['AverageNs'])/1e3:.1f} | {m['Calls']}" if m else "— | —") + " |")
    w("\nThe rocprofv3 column averages over every call of the process (warm-up, timed steps, the profiled steps, the phase measurements), so the shape mix differs\n"
      "slightly from one train_step; HIP events add ≈ 1.5 µs to a kernel of a few µs.\n")
pt = os.path.join(P, R + "_pmc_summary.json")
if os.path.exists(pt):
    w("## PMC passes over ONE policy phase alone (rollout + estimator + BPTT): the tall kernels\n")
    w("| kernel | launches | avg µs | read MB/launch | write MB/launch | MFMA pipe busy | clock GHz |\n|---|---|---|---|---|---|---|")
    for k, v in list(json.load(open(pt)).items())[:8]:
        if "launches_mfma" in v:
            w(f"| `{k[:72]}` | {v.get('launches_mfma','')} | {v.get('avg_us_mfma',0):.1f} | {v.get('fetch_bytes_per_launch_corrected',0)/1e6:.1f} | {v.get('write_bytes_per_launch',0)/1e6:.1f} | "
              f"{100*v.get('mfma_util',0):.1f} % | {v.get('eff_clock_GHz',0):.2f} |")
    w("")
w("## PMC passes over ONE 256-row reverse loop alone (FETCH_SIZE x 1 KiB x 2 = corrected read bytes, WRITE_SIZE x 1 KiB; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs ÷ GRBM_GUI_ACTIVE / 8 XCDs)\n")
w("| kernel | launches | avg µs | read MB/launch | write MB/launch | MFMA pipe busy (GRBM window) | MFMA pipe busy over the kernel's own duration at 2.4 GHz (lower bound) |\n|---|---|---|---|---|---|---|")
for k, v in pmc.items():
    if "launches_mfma" not in v:
        continue
    w(f"| `{k[:72]}` | {v.get('launches_mfma','')} | {v.get('avg_us_mfma',0):.1f} | {v.get('fetch_bytes_per_launch_corrected',0)/1e6:.2f} | {v.get('write_bytes_per_launch',0)/1e6:.2f} | "
      f"{100*v.get('mfma_util',0):.1f} % | {100*v.get('mfma_util_kernel_window_lower_bound',0):.1f} % |")
w("\n(GRBM_GUI_ACTIVE spans more than the kernel on dispatches of a few µs - the reduce script's effective clock reads 3.6-12 GHz there - which dilutes the first utilisation\n"
  "column by the same factor; the second divides the busy cycles by the kernel's own duration at the maximum clock and is a lower bound.)  (The effective-clock figure of the reduce script is meaningless on dispatches of a few µs and is not shown.)  The chain kernels are latency-bound, not bandwidth- or\n"
  "MFMA-bound: a 256 x 512 x 512 product split four ways over K is two 64-deep stages per workgroup.\n")
oc = bench.get("other_configs", {})
if oc:
    w("## Other BASELINE configs at full size (same build, same process as the bench line, outside its timed region)\n")
    for k, d in oc.items():
        if d and "ms_per_step" in d and "steps_per_s" not in d:      # the estimator-on pixel step: a cost statement, not a config of BASELINE.json
            w(f"* {k}: {d['ms_per_step']:.0f} ms/step ({d.get('workload', '')}; {d.get('images_decoded', 0)} decoded images, library-counted {d.get('gemm_flop', 0)/1e12:.1f} TFLOP = "
              f"{100*(d.get('frac') or 0):.1f} % of the fp32-MFMA peak)")
            continue
        if d and "ms_per_step" in d:
            w(f"* {k}: {d['ms_per_step']:.1f} ms/step = {d['steps_per_s']:.2f} steps/s, {d['algorithmic_tflops']:.1f} algorithmic TFLOP/s = {100*d['frac_of_f32_mfma_peak']:.1f} % of the fp32-MFMA peak")
            php = d.get("roofline_phases") or {}
            if php and "error" not in php:
                w("  * phases alone (ms, % of the relevant peak): " + "; ".join(f"`{n}` {v['ms']:.2f} ms, {100*v['frac']:.1f} %" for n, v in php.items()))
    pe = oc.get("pixel_epistemic_f1")
    if pe:
        w(f"* pixel_epistemic_f1 (SURVEY 8(f)-1, REDUCED size: {pe['config']['belief_rows']} belief rows, S = {pe['config']['ambiguity_samples']}): "
          + "; ".join(f"`{k}` {v['ms']:.1f} ms, {v['achieved']:.1f} TFLOP/s = {100*v['frac']:.1f} % (library-counted GEMM FLOPs)" for k, v in pe.items() if k != "config"))
    w("")
sb = bench.get("small_batch", {})
if sb:
    w("## Small-batch reverse loops (hipGraph replay; bound by streaming the score network's weights once per denoise step)\n")
    for k, d in sb.items():
        w(f"* {k}: {d['rows']} rows, {d['us_per_denoise_step']:.1f} µs per denoise step = {d['denoise_steps_per_s']:.0f} steps/s; "
          f"{d['weight_bytes_per_step']/1e6:.1f} MB of weights per step -> {d['achieved_GBps']:.0f} GB/s = {100*d['achieved_GBps']/d['peak_GBps']:.1f} % of 8 TB/s")
    w("")
open(os.path.join(P, "README.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out)[:2500])
