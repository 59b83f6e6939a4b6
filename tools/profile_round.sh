#!/bin/bash
# Full measurement checkpoint on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes.
mkdir -p gpurun_out
timeout -k 10 400 python bench.py 2>gpurun_out/bench_final.err | tail -1 > gpurun_out/bench_final.json || exit 1
cut -c1-160 gpurun_out/bench_final.json
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_final
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1) || exit 1
echo "kernel-trace done"
bash tools/pmc_bench.sh
