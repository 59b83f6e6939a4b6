#!/bin/bash
# PMC passes over tools/bench_fbwd.py (field backward alone).  usage: pmc_fbwd.sh <tag> [env...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp
run() { t=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmcf_${tag}_$t -- python $R/tools/bench_fbwd.py > $R/gpurun_out/pmcf_${tag}_$t.log 2>&1
  echo "pmc $tag $t rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAVES
python3 - <<PY
import csv, glob, collections, re
for t in ("sq1", "sq2"):
    for f in glob.glob("$R/gpurun_out/pmcf_${tag}_%s/**/*counter_collection.csv" % t, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()[:44]
            acc[(name, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (n, c, d), v in acc.items():
            per[n][c].append(v)
        for n, cs in per.items():
            if "field_bwd" not in n: continue
            print(n, {c: round(sorted(v)[len(v)//2]) for c, v in cs.items()})
PY
