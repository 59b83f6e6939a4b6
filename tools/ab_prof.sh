#!/bin/bash
# ab_prof.sh : rocprofv3 kernel stats of the bench step with the in-tree library and with every tools/_alt/lib*.so (GPU box)
export TMPDIR=/tmp
for l in "" tools/_alt/lib*.so; do
  name=$(basename "${l:-intree}" .so); OUT=$GRAFT_REPO_ROOT/gpurun_out/abprof_$name
  export UMHS_LIB_PATH=${l:+$GRAFT_REPO_ROOT/$l}
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT.log 2>&1) || exit 1
  echo "== $name"; f=$(find $OUT -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -24
done
