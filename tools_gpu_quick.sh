#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$1" > gpurun_out/testq.log 2>&1
tail -5 gpurun_out/testq.log; grep -E "AssertionError:" gpurun_out/testq.log | head
