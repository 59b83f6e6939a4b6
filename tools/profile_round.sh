#!/bin/bash
# Measurement checkpoint on the GPU box, in two gpurun calls (each fits the 20-minute limit):
#   bash tools/profile_round.sh lines   bench lines (C2 headline with sampler_step + cpu_baseline, C3, C5) + rocprofv3 kernel stats of the same commands
#   bash tools/profile_round.sh pmc     PMC passes for C2, C3, C5 (tools/pmc_bench.sh)
# Results land in gpurun_out/; tools/collect_profiles.sh copies the summaries into profiles/rNN/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "${1:-lines}" = pmc ]; then
  bash tools/pmc_bench.sh C2 C3 C5
  exit $?
fi
timeout -k 10 400 python bench.py 2>gpurun_out/bench_C2.err | tail -1 > gpurun_out/bench_C2.json || exit 1
cut -c1-200 gpurun_out/bench_C2.json
for c in C3 C5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-sampler-step --no-eval-image 2>gpurun_out/bench_$c.err | tail -1 > gpurun_out/bench_$c.json || exit 1
  cut -c1-160 gpurun_out/bench_$c.json
done
for c in C2 C3 C5; do
  OUT=$R/gpurun_out/prof_$c
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-sampler-step --no-other-configs --no-eval-image > $R/gpurun_out/prof_$c.log 2>&1) || exit 1
  echo "kernel-trace $c done"
done
# the eval-image path (tools/bench_eval.py: 300 training steps, then get_outputs_for_camera_ray_bundle on a 256 x 256 camera)
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eval -- python $R/tools/bench_eval.py > $R/gpurun_out/prof_eval.log 2>&1) || exit 1
echo "kernel-trace eval done: $(grep 'eval image' gpurun_out/prof_eval.log)"

