#!/bin/bash
source tools/gpu_call.sh
step 600 cur.log python -m pytest tests/test_device_ops_gpu.py tests/test_hip_parity.py -x -q
tail -3 gpurun_out/cur.log
bash tools/vb.sh "-" "65536 16384"
step 300 kstats65.log bash tools/kstats.sh r03_65536 65536
head -7 gpurun_out/kstats65.log
