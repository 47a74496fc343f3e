#!/bin/bash
source tools/gpu_call.sh
step 200 kb12_all.json python tools/kbench.py 65536
step 200 kb12_all_8192.json python tools/kbench.py 8192
cat gpurun_out/kb12_*.json
step 900 r3_t12.log python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider -k "dynamics or kernels or wave"
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t12.log | cut -c1-400
