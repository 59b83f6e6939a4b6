#!/bin/bash
# timing-only ablations of the field backward (results are wrong by construction)
set -e
cd $GRAFT_REPO_ROOT/unsupervised-hyperspectral-nerf_amd
CS=csrc; INC=../include
build() { # name flags
  hipcc --offload-arch=gfx950 -O3 -fPIC -munsafe-fp-atomics -std=c++17 -I$INC -I$CS $2 -c $CS/umhs_field.hip -o /tmp/f_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC $CS/umhs_kernels.o /tmp/f_$1.o -o /tmp/lib_$1.so
}
build base ""
build nodw "-DUMHS_ABL_NO_DW"
build nosync "-DUMHS_ABL_NO_SYNC"
build nodw_nosync "-DUMHS_ABL_NO_DW -DUMHS_ABL_NO_SYNC"
cd $GRAFT_REPO_ROOT
for v in base nodw nosync nodw_nosync; do
  cp /tmp/lib_$v.so unsupervised-hyperspectral-nerf_amd/umhsnerf/libumhs_hip.so
  echo "== $v"; ONLY=1 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "N="
done
