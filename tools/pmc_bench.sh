#!/bin/bash
# rocprofv3 PMC passes over bench.py (separate runs per counter group, as MI355X_MICROARCH.md prescribes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmcb_$tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmcb_$tag.log 2>&1
  echo "pmc $tag rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
run grbm GRBM_GUI_ACTIVE
