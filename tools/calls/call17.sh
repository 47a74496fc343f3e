#!/bin/bash
source tools/gpu_call.sh
step 1100 r3_t17.log python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t17.log | cut -c1-400
bash tools/vb.sh "-" "65536 8192 16384"
