#!/bin/bash
# Socket power and shader clock sampled (rocm-smi, every ~0.3 s) while a phase runs in a loop: is the chip at its power cap, and at which clock?
# usage: bash tools/perf/power_probe.sh <phase_profile phase | step> [reps]
cd "$GRAFT_REPO_ROOT"
what=$1; reps=${2:-400}
if [ "$what" = step ]; then python bench.py --steps $reps --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1 &
else python tools/perf/phase_profile.py $what $reps > /dev/null 2>&1 & fi
pid=$!
sleep 25
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|Average Graphics Package Power|sclk clock level" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 0.3
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
