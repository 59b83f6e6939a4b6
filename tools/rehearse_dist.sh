#!/bin/bash
# 2 ranks on ONE GPU over gloo: the N>1 code path of bench.py end to end (per-level-group reduction overlapped with the backward, exchange diagnostics)
UMHS_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 2>gpurun_out/rehearse.err | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['sanity']['loss'], d['dist'])"
