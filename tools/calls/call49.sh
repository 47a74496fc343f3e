#!/bin/bash
source tools/gpu_call.sh
step 1100 gpu_all.log python -m pytest tests -x -q -m gpu
tail -5 gpurun_out/gpu_all.log
bash tools/vb.sh "-" "65536 8192"
