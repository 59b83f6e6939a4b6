#!/bin/bash
# PMC passes over tools/bench_field.py (field + hash kernels at N=262144).  Counters in separate runs (slot limits).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|TCC_[A-Z_0-9a-z\[\]]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy" | sort -u > $R/gpurun_out/pmc_list.txt
wc -l $R/gpurun_out/pmc_list.txt
run() { # tag counters...
  tag=$1; shift
  ONLY=1 timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/tools/bench_field.py > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAVES
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
find $R/gpurun_out -name "*counter_collection.csv" | head
