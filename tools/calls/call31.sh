#!/bin/bash
source tools/gpu_call.sh
step 300 occ2_agree.log env PARC_ENV_LIB=variants/libparc_env_occ2.so python -m pytest tests/test_dynamics_gpu.py -x -q -k "kernels_agree"
bash tools/vb.sh "- variants/libparc_env_occ1.so variants/libparc_env_occ2.so" "65536 8192"
