#!/bin/bash
# rocprofv3 PMC passes over bench.py (separate runs per counter group, as MI355X_MICROARCH.md prescribes)
#   usage: bash tools/pmc_bench.sh [C2|C3|C5 ...]   -> gpurun_out/pmcb_<config>_<group>/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { cfg=$1; tag=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmcb_${cfg}_$tag -- python $R/bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-sampler-step --no-other-configs --no-eval-image > $R/gpurun_out/pmcb_${cfg}_$tag.log 2>&1
  rc=$?; echo "pmc $cfg $tag rc=$rc"; return $rc; }
for cfg in ${@:-C2}; do
  run $cfg fetch FETCH_SIZE &&
  run $cfg write WRITE_SIZE &&
  run $cfg sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT &&
  run $cfg grbm GRBM_GUI_ACTIVE || exit 1
done
