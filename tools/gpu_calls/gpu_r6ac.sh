#!/bin/bash
# round 6 call ac: bench.py under world_size 2 as the driver launches it (torch.distributed.run), every rank on cuda:0 with gloo for the
# collectives (--_share-gpu: test hook) - the real workloads, the cfg-4 side figure and the rank-0 report under N > 1 on the one-GPU box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
P=29611
for ARGS in "" "--config cfg4" "--config cfg4 --collective allreduce" "--config cfg5 --no-side" "--config apitrue --no-side" "--config cfg3 --no-side"; do
  P=$((P+1))
  echo "== bench.py --gpus 2 $ARGS"
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 3 --warmup 1 --_share-gpu $ARGS 2>&1 | grep -v "amdgpu.ids\|^W\|^\*\*\*\|OMP_NUM" | tail -3 | cut -c1-900
done | tee gpurun_out/r6ac_bench_two_ranks_shared_gpu.log
