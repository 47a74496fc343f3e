#!/bin/bash
source tools/gpu_call.sh
PARC_ENV_LIB=variants/libparc_env_counts.so step 300 counts27.log python tools/wave_stamps.py 65536
cat gpurun_out/counts27.log.err | grep -v amdgpu.ids | tail -18
