#!/bin/bash
# sweep_env.sh VAR "v1 v2 ..." <cmd...> : run the command once per value of the environment variable, twice round-robin (GPU box)
var=$1; vals=$2; shift 2
for rep in 1 2; do
  for v in $vals; do
    echo "== $var=$v (rep $rep)"
    env $var=$v timeout -k 10 300 "$@" 2>&1 | grep -E "N=|\[bench\] gpu"
  done
done
