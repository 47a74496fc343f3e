#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "variants/libparc_env_gfac.so -" "65536 8192"
