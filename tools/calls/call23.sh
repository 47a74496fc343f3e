#!/bin/bash
source tools/gpu_call.sh
step 600 r3_t23.log python -m pytest tests -m gpu -q -p no:cacheprovider -k "soak" --durations=3
grep "^E  .*Error\|^FAILED\|passed\|failed\|s call" gpurun_out/r3_t23.log | cut -c1-300
step 300 r3_bench23.json python bench.py --steps 50 --warmup 10 --cpu-seconds 6
tail -c 900 gpurun_out/r3_bench23.json
