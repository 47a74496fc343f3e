#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "-" "65536 8192"
step 600 r3_t18.log python -m pytest tests -m gpu -q -p no:cacheprovider -k "kernels_agree or wave_kernel or cpu_build or bench_sizes"
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t18.log | cut -c1-300
