#!/bin/bash
# A/B two builds of libaid_hip.so on the bench's two timed regions, alternating in ONE gpurun call (boxes differ by ~10 %): usage  ab_bench.sh <other.so> [reps]
OTHER=$1; REPS=${2:-2}
export AID_BENCH_SKIP=phases,graphs,denoise
line() { python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'elided', d['ms_per_step_dead_belief_elided'])"; }
for rep in $(seq $REPS); do
  echo "== default"; line
  echo "== $OTHER"; AID_HIP_LIB=$OTHER line
done
