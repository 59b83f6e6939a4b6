#!/bin/bash
# In-step A/B on ONE GPU box: alternates environment settings over the same bench command (box-to-box spread is 1.5-3 %, so only
# numbers from one gpurun call compare; kernels that are faster alone are not always faster in the step).
#   usage: bash tools/ab_step.sh <config> <reps> "ENV_A=.. ENV_B=.." "ENV_A=.. ENV_B=.." ...
#   e.g.   bash tools/ab_step.sh C3 3 "UMHS_SIDE_STREAM=1" "UMHS_SIDE_STREAM=0"      (stderr of a failing run: gpurun_out/ab_step.err)
cfg=$1; reps=$2; shift 2
for i in $(seq 1 $reps); do
  for setting in "$@"; do
    echo -n "$cfg [$setting]: "
    env $setting timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --steps 50 --warmup 10 2>gpurun_out/ab_step.err | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mean', d['ms_per_step'], 'median', d['ms_per_step_median'], 'ms')" || { tail -5 gpurun_out/ab_step.err; exit 1; }
  done
done
