#!/bin/bash
source tools/gpu_call.sh
step 600 post4_parity.log python -m pytest tests/test_hip_parity.py tests/test_device_ops_gpu.py -x -q
tail -3 gpurun_out/post4_parity.log
bash tools/vb.sh "-" "65536 8192"
step 300 pmc_post.log bash tools/pmc2.sh post "k_env_post<0" 65536 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS"
grep -A8 avg_per gpurun_out/pmc_post.log
