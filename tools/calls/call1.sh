#!/bin/bash
source tools/gpu_call.sh
step 1100 r3_t1.log python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider
tail -5 gpurun_out/r3_t1.log
step 300 r3_cpu_gpu_diag.log python tests/diag_dyn_cpu_gpu.py 2048
step 300 r3_settle.log python tools/dyn_settle_diag.py 16384 gpurun_out/r3_settle.json
tail -15 gpurun_out/r3_settle.log
step 300 r3_bench.json python bench.py --steps 200 --warmup 20
step 200 r3_bench_8192.json python bench.py --envs 8192 --steps 300 --no-cpu-baseline
step 300 r3_bench_ppo.json python bench.py --ppo 1 --steps 96 --warmup 32 --no-cpu-baseline
tail -c 600 gpurun_out/r3_bench.json
