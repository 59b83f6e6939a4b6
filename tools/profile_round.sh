#!/bin/bash
# Full measurement checkpoint on the GPU box: bench lines (C2 headline + C3 + C5), rocprofv3 kernel stats of the same commands,
# PMC passes (tools/pmc_bench.sh).  Results land in gpurun_out/; tools/collect_profiles.sh copies the summaries into profiles/rNN/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py 2>gpurun_out/bench_C2.err | tail -1 > gpurun_out/bench_C2.json || exit 1
cut -c1-200 gpurun_out/bench_C2.json
for c in C3 C5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline 2>gpurun_out/bench_$c.err | tail -1 > gpurun_out/bench_$c.json || exit 1
  cut -c1-160 gpurun_out/bench_$c.json
done
for c in C2 C3; do
  OUT=$R/gpurun_out/prof_$c
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_$c.log 2>&1) || exit 1
  echo "kernel-trace $c done"
done
bash tools/pmc_bench.sh
