#!/bin/bash
source tools/gpu_call.sh
PARC_DYN_SEGMENTS=none step 200 kb5_none.json python tools/kbench.py 65536
PARC_DYN_SEGMENTS=capsules step 200 kb5_caps.json python tools/kbench.py 65536
step 200 kb5_all.json python tools/kbench.py 65536
step 200 kb5_all_8192.json python tools/kbench.py 8192
cat gpurun_out/kb5_*.json
step 900 r3_t5.log python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider -k "dynamics or kernels or wave"
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t5.log | cut -c1-400
