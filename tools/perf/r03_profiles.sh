#!/bin/bash
# Round-3 rocprofv3 passes (GPU box, from the repo root).  Kernel stats of the bench command in both schedules, kernel stats per phase
# (tools/perf/phase_profile.py), and the PMC passes for ONE 256-row reverse loop alone: the kernels of the concurrent chains
# (gemm_mid_kernel<1,*>, adaln_sum_kernel<2,4>, adaln_lp_kernel<2>, head2_slab_p_sample_kernel) - counter collection on this pool does not
# survive concurrent streams, and one chain alone runs exactly the kernels, shapes and launch geometry each of the three chains runs.
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r03; rm -rf $OUT; mkdir -p $OUT
P=tools/perf/phase_profile.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_noov -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-overlap > $OUT/stats_noov.log 2>&1
echo stats_noov done
# "_cold": no warm-up train_steps in the process, so that a phase's summary holds the phase's kernels only (5 calls + 1 for the launch dump)
for ph in elbo rollout_bptt efe chain1 chains3 value_dynamics optimiser; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/phase_$ph -- python3 $P ${ph}_cold 5 > $OUT/phase_$ph.log 2>&1
  echo phase $ph done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/phase_pixel_epistemic -- python3 $P pixel_epistemic 1 > $OUT/phase_pixel_epistemic.log 2>&1
echo phase pixel_epistemic done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $P chain1_direct_cold 1 > $OUT/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $P chain1_direct_cold 1 > $OUT/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $P chain1_direct_cold 1 > $OUT/mfma.log 2>&1
echo mfma done
python3 tools/perf/pmc_reduce.py $OUT
python3 tools/perf/r03_collect.py $OUT
