#!/bin/bash
source tools/gpu_call.sh
step 600 cur.log python -m pytest tests/test_device_ops_gpu.py tests/test_hip_parity.py -x -q
tail -5 gpurun_out/cur.log
bash tools/vb.sh "-" "65536 8192"
