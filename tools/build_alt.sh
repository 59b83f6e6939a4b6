#!/bin/bash
# build_alt.sh <name> [file.hip replacing the same-named source in csrc ...]  ->  tools/_alt/lib<name>.so  (A/B experiments)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/unsupervised-hyperspectral-nerf_amd/csrc; OUT=$ROOT/tools/_alt; name=$1; shift
mkdir -p $OUT/obj_$name
objs=""
for src in umhs_kernels umhs_field umhs_sampler umhs_data umhs_metrics; do
  f=$CS/$src.hip
  for alt in "$@"; do [ "$(basename $alt)" = "$src.hip" ] && f=$alt; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -munsafe-fp-atomics -std=c++17 $EXTRA -I$ROOT/include -I$CS -c $f -o $OUT/obj_$name/$src.o &
  objs="$objs $OUT/obj_$name/$src.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/lib$name.so
echo built $OUT/lib$name.so
