#!/bin/bash
# round 5 call aq: the driver's launch forms on the shipped build: plain python and torch.distributed.run with one rank
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-side 2> gpurun_out/r5aq.err | grep '^{' | cut -c1-420 | tee gpurun_out/r5aq_torchrun1.json
echo "rc=${PIPESTATUS[0]}"
timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c1-300
timeout 300 python bench.py --config cfg5 --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c1-300
