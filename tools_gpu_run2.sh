#!/bin/bash
# GPU session: parity tests, smoke, bench, rocprof kernel trace
mkdir -p gpurun_out
if [ "$SKIP_TESTS" != "1" ]; then
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/test2.log 2>&1
rc=$?; tail -4 gpurun_out/test2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed rc=$rc"; exit $rc; fi
fi
nproc; python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2>&1 | tee gpurun_out/bench1.log || { exit 1; }
tail -2 gpurun_out/bench1.log
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof1
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
echo "rocprof rc=$?"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof1.log
find $OUT -name "*stats*" | head
