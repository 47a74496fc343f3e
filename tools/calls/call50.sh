#!/bin/bash
source tools/gpu_call.sh
step 300 hbm_copy.json python tools/hbm_copy_bw.py
cat gpurun_out/hbm_copy.json
