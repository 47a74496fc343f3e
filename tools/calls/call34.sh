#!/bin/bash
source tools/gpu_call.sh
step 600 post4_parity.log python -m pytest tests/test_hip_parity.py tests/test_device_ops_gpu.py -x -q
tail -3 gpurun_out/post4_parity.log
bash tools/vb.sh "-" "65536 8192"
