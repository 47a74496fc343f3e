#!/bin/bash
source tools/gpu_call.sh
step 600 cur.log python -m pytest tests/test_device_ops_gpu.py tests/test_hip_parity.py -x -q
tail -3 gpurun_out/cur.log
bash tools/vb.sh "-" "8192 16384"
PARC_CURRICULUM_TWO_LAUNCHES=1 bash tools/vb.sh "-" "8192 16384"
step 300 kstats_r03.log bash tools/kstats.sh r03_8192 8192
head -8 gpurun_out/kstats_r03.log
