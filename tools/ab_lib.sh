#!/bin/bash
# ab_lib.sh <cmd...> : run the command alternately with every tools/_alt/lib*.so and the in-tree library (GPU box)
for rep in 1 2; do
  for l in "" tools/_alt/lib*.so; do
    echo "== ${l:-in-tree} (rep $rep)"
    UMHS_LIB_PATH=${l:+$PWD/$l} timeout -k 10 300 "$@" 2>&1 | grep -E "N=|\[bench\] gpu"
  done
done
