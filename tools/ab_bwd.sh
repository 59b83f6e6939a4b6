#!/bin/bash
for v in 0 1 0 1; do echo "== bwd variant $v"; UMHS_BWD_VARIANT=$v ONLY=1 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep "N="; done
UMHS_BWD_VARIANT=1 timeout -k 10 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "field_bwd or end_to_end or golden or train_iteration" 2>&1 | tail -2
