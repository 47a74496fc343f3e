#!/bin/bash
source tools/gpu_call.sh
export PARC_BENCH_SHARE_GPU=1
step 400 torchrun2.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5
grep '^{' gpurun_out/torchrun2.log | cut -c1-400; tail -3 gpurun_out/torchrun2.log.err
unset PARC_BENCH_SHARE_GPU
step 400 torchrun1.log python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 20 --warmup 5 --ppo 1 --no-cpu-baseline
grep '^{' gpurun_out/torchrun1.log | cut -c1-300; tail -3 gpurun_out/torchrun1.log.err
