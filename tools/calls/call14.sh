#!/bin/bash
source tools/gpu_call.sh
step 600 r3_t14.log python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider -k "fused_curriculum or large_library or never_done or reset_done or graph_step or golden"
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t14.log | cut -c1-400
step 200 kb14_all.json python tools/kbench.py 65536
step 200 kb14_all_8192.json python tools/kbench.py 8192
PARC_SPLIT_TAIL=1 step 200 kb14_split_8192.json python tools/kbench.py 8192
PARC_SPLIT_TAIL=1 step 200 kb14_split.json python tools/kbench.py 65536
cat gpurun_out/kb14_*.json
