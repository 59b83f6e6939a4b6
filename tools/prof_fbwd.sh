#!/bin/bash
# rocprofv3 kernel stats of tools/bench_fbwd.py (field backward alone); usage: prof_fbwd.sh <tag>   (env CASES / UMHS_BWD_TF pass through)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/proff_$tag -- python $R/tools/bench_fbwd.py > $R/gpurun_out/proff_$tag.log 2>&1)
f=$(find $R/gpurun_out/proff_$tag -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print("  %-60s %5s calls avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
