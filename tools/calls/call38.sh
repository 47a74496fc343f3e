#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "variants/libparc_env_epb32.so" "8192 4096"
bash tools/vb.sh "-" "8192 4096"
