#!/bin/bash
source tools/gpu_call.sh
step 1100 prof_r03.log bash tools/profile_round.sh r03
tail -3 gpurun_out/prof_r03.log
# the profiles must be in place for the bench lines to carry the counter-derived fields of THIS build
python3 tools/publish_profiles.py r03 r03 r03_8192 > gpurun_out/publish_tmp.log 2>&1 || tail -5 gpurun_out/publish_tmp.log
step 900 bset_r03.log bash tools/bench_set.sh r03
cat gpurun_out/bset_r03.log
step 300 kstats_r03.log bash tools/kstats.sh r03_8192 8192
step 300 bench_ppo_r03.json python bench.py --ppo 1 --steps 96 --warmup 32 --no-cpu-baseline
step 300 bench_nosegments_r03.log env PARC_DYN_SEGMENTS=none python bench.py --steps 200 --warmup 20 --no-cpu-baseline
