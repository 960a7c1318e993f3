#!/bin/bash
# A/B an environment switch of bench.py (e.g. AID_ROWLN_DEBUG=37) on the two timed regions, alternating in ONE gpurun call: usage  ab_env.sh VAR=value [reps]
SW=$1; REPS=${2:-3}
export AID_BENCH_SKIP=phases,graphs,denoise,x6
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'elided', d['ms_per_step_dead_belief_elided'])"; }
for rep in $(seq $REPS); do
  echo "== default"; python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | show
  echo "== $SW"; env $SW python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | show
done
