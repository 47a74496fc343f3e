#!/bin/bash
source tools/gpu_call.sh
step 900 bset_r03.log bash tools/bench_set.sh r03
cat gpurun_out/bset_r03.log
step 300 kstats_r03.log bash tools/kstats.sh r03_8192 8192
cat gpurun_out/kstats_r03.log
step 300 bench_ppo_r03.json python bench.py --ppo 1 --steps 96 --warmup 32 --no-cpu-baseline
step 300 bench_nosegments_r03.log env PARC_DYN_SEGMENTS=none python bench.py --steps 200 --warmup 20 --no-cpu-baseline
