#!/bin/bash
source tools/gpu_call.sh
step 1100 r3_t10.log python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t10.log | cut -c1-400
step 300 nan_hunt10.log python tools/nan_hunt.py 16384 1024
grep -v "Loading\|Synthetic" gpurun_out/nan_hunt10.log | cut -c1-200 | head -5
step 300 nan_hunt10b.log python tools/nan_hunt.py 65536 0
grep -v "Loading\|Synthetic" gpurun_out/nan_hunt10b.log | cut -c1-200 | head -5
step 300 r3_replay10.log python tools/replay_diag.py 32
grep "steps_total" gpurun_out/r3_replay10.log | cut -c1-1000
SETTLE_STEPS=240 step 300 r3_settle10.log python tools/dyn_settle_diag.py 16384
grep "t=4.0s\|t=8.0s" gpurun_out/r3_settle10.log
step 200 kb10_all.json python tools/kbench.py 65536
step 200 kb10_all_8192.json python tools/kbench.py 8192
cat gpurun_out/kb10_*.json
