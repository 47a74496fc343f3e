#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "- variants/libparc_env_nocull.so - variants/libparc_env_nocull.so" "65536" 
bash tools/vb.sh "- variants/libparc_env_nocull.so" "8192"
PARC_ENV_LIB=variants/libparc_env_nocull.so PARC_SKIP_FLAG_CHECK=1 step 600 r3_t16.log python -m pytest tests -m gpu -q -p no:cacheprovider -k "kernels_agree or wave_kernel or cpu_build"
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t16.log | cut -c1-300
