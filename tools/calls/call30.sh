#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "- variants/libparc_env_lb512.so" "65536 8192"
