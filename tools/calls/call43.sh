#!/bin/bash
source tools/gpu_call.sh
step 600 soak_final.log python tools/soak.py
tail -10 gpurun_out/soak_final.log
