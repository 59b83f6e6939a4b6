#!/bin/bash
for cfg in "0 0" "1 0" "1 1" "1 2" "1 3" "1 4"; do set -- $cfg; echo "== variant $1 stagger $2"; UMHS_FWD_VARIANT=$1 UMHS_FWD_STAGGER=$2 ONLY=1 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "N=" | cut -c1-60; done
