#!/bin/bash
source tools/gpu_call.sh
step 1100 r3_t22.log python -m pytest tests -m gpu -q -p no:cacheprovider --durations=4
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t22.log | cut -c1-300
bash tools/vb.sh "-" "65536 8192"
