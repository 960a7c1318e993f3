#!/bin/bash
# A/B two builds of libaid_hip.so in one gpurun call (boxes differ by ~10 %): usage  ab_lib.sh <other.so> [script args...]
# e.g. make -C active-inference-diffusion_amd/csrc BUILD=build_nt EXTRA=-DAID_PANEL_B_AUX=2 OUT=../aid_amd/lib/libaid_hip_nt.so
OTHER=$1; shift
for rep in 1 2; do
  echo "== default"; python tools/perf/denoise_rate.py "$@" 2>/dev/null | tail -3
  echo "== $OTHER"; AID_HIP_LIB=$OTHER python tools/perf/denoise_rate.py "$@" 2>/dev/null | tail -3
done
echo "== default train_step"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['denoise_steps_per_s'])"
echo "== $OTHER train_step"; AID_HIP_LIB=$OTHER python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['denoise_steps_per_s'])"
