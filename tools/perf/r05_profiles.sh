#!/bin/bash
# Round-5 rocprofv3 passes (GPU box, from the repo root; ~12 min).  Kernel stats of the bench command in both schedules, kernel stats per phase
# (tools/perf/phase_profile.py), of one cfg-3 / cfg-4 process (tools/perf/cfg{3,4}_profile.py, both schedules) and of the small-batch loops; PMC passes
# (separate runs, --kernel-trace only beside them) for ONE 256-row reverse loop alone - the kernels of the concurrent chains - and for one
# policy phase alone - the estimator's tall full-row kernels (counter collection on this pool does not survive concurrent streams).
# usage: r05_profiles.sh [part ...]   parts: stats phases cfg34 small pmc_chain pmc_tall x6   (default: all but x6)
# x6: the policy phase (rollout + estimator + BPTT) with the opt-in six-term bf16 arithmetic (AID_GEMM_X6=1): kernel stats + the MFMA-busy / clock pass
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r05; mkdir -p $OUT
P=tools/perf/phase_profile.py
PARTS="${@:-stats phases cfg34 small pmc_chain pmc_tall}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
RP="rocprofv3 --kernel-trace --stats --output-format csv"
if has stats; then
  # AID_BENCH_SKIP=phases,denoise,graphs: the traced process runs the two timed regions (+ warm-up, + the two single-stream profiled passes), not the untimed
  # per-phase / denoise-only / small-batch measurements, so that a kernel's share of kernel time is its share in the schedules the line times
  export AID_BENCH_SKIP=phases,denoise,graphs,x6
  $RP -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1; echo stats done
  $RP -d $OUT/stats_noov -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-overlap > $OUT/stats_noov.log 2>&1; echo stats_noov done
  unset AID_BENCH_SKIP
fi
if has phases; then
  # "_cold": no warm-up train_steps in the process, so that a phase's summary holds the phase's kernels only
  for ph in elbo rollout_bptt efe chain1 chains3 chains2 value_dynamics optimiser; do
    $RP -d $OUT/phase_$ph -- python3 $P ${ph}_cold 5 > $OUT/phase_$ph.log 2>&1; echo phase $ph done
  done
  $RP -d $OUT/phase_pixel_epistemic -- python3 $P pixel_epistemic 1 > $OUT/phase_pixel_epistemic.log 2>&1; echo phase pixel_epistemic done
fi
if has cfg34; then
  $RP -d $OUT/cfg4 -- python3 tools/perf/cfg4_profile.py 5 > $OUT/cfg4.log 2>&1; echo cfg4 done
  $RP -d $OUT/cfg3 -- python3 tools/perf/cfg3_profile.py 5 > $OUT/cfg3.log 2>&1; echo cfg3 done
  export AID_NO_OVERLAP=1
  $RP -d $OUT/cfg4_noov -- python3 tools/perf/cfg4_profile.py 5 > $OUT/cfg4_noov.log 2>&1; echo cfg4_noov done
  AID_DUMP=$OUT/cfg4_launches.csv python3 tools/perf/cfg4_profile.py 2 > $OUT/cfg4_dump.log 2>&1; echo cfg4 launches dumped
  $RP -d $OUT/cfg3_noov -- python3 tools/perf/cfg3_profile.py 5 > $OUT/cfg3_noov.log 2>&1; echo cfg3_noov done
  unset AID_NO_OVERLAP
fi
if has small; then
  for b in 1 32; do
    $RP -d $OUT/small_b$b -- python3 tools/perf/small_batch_prof.py $b > $OUT/small_b$b.log 2>&1; echo small_b$b done
  done
fi
pmc() {   # pmc <tag> <phase>: three counter passes over one call of the phase
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$1/fetch -- python3 $P $2 1 > $OUT/$1_fetch.log 2>&1; echo $1 fetch done
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$1/write -- python3 $P $2 1 > $OUT/$1_write.log 2>&1; echo $1 write done
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$1/mfma -- python3 $P $2 1 > $OUT/$1_mfma.log 2>&1; echo $1 mfma done
  python3 tools/perf/pmc_reduce.py $OUT/$1 > $OUT/$1_reduce.log 2>&1
}
if has x6; then
  export AID_GEMM_X6=1
  $RP -d $OUT/phase_efe_x6 -- python3 $P efe_cold 5 > $OUT/phase_efe_x6.log 2>&1; echo phase efe x6 done
  mkdir -p $OUT/pmc_tall_x6; pmc pmc_tall_x6 efe_cold
  unset AID_GEMM_X6
  mkdir -p $OUT/collect
  f=$(find $OUT/phase_efe_x6 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/collect/r05_phase_efe_x6_kernel_stats.csv
  [ -f $OUT/pmc_tall_x6/pmc_summary.json ] && cp $OUT/pmc_tall_x6/pmc_summary.json $OUT/collect/r05_pmc_x6_summary.json
fi
if has pmc_chain; then mkdir -p $OUT/pmc_chain; pmc pmc_chain chain1_direct_cold; fi
if has pmc_tall; then mkdir -p $OUT/pmc_tall; pmc pmc_tall efe_cold; fi
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
python3 tools/perf/r05_collect.py $OUT
f=$(find $OUT/cfg4_noov -name '*kernel_stats.csv' | head -1)
if [ -f "$OUT/cfg4_launches.csv" ] && [ -n "$f" ]; then python3 tools/perf/cfg4_roofline.py $OUT/cfg4_launches.csv $f 8 > $OUT/collect/r05_cfg4_roofline.md; cp $OUT/cfg4_launches.csv $OUT/collect/r05_cfg4_launches.csv; fi
