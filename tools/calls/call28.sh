#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "- variants/libparc_env_tp.so - variants/libparc_env_tp.so" "65536"
bash tools/vb.sh "- variants/libparc_env_tp.so" "8192"
PARC_ENV_LIB=variants/libparc_env_tp.so step 600 r3_t28.log python -m pytest tests -m gpu -q -p no:cacheprovider -k "kernels_agree or wave_kernel or cpu_build or bench_sizes or soak"
grep "^E  .*\|^FAILED\|passed\|failed" gpurun_out/r3_t28.log | cut -c1-300
