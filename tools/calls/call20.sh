#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "-" "65536 8192"
step 1100 r3_t20.log python -m pytest tests -m gpu -q -p no:cacheprovider
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t20.log | cut -c1-300
