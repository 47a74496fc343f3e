#!/bin/bash
source tools/gpu_call.sh
PARC_ENV_LIB=variants/libparc_env_tl.so step 300 tl11.log python tools/wave_timeline.py 65536
grep -A40 "sample 2" gpurun_out/tl11.log
PARC_ENV_LIB=variants/libparc_env_stamps.so step 300 stamps11.log python tools/wave_stamps.py 65536
tail -60 gpurun_out/stamps11.log | head -75; cat gpurun_out/stamps11.log.err | tail -20
