#!/bin/bash
source tools/gpu_call.sh
step 600 r3_t26.log python -m pytest tests -m gpu -q -p no:cacheprovider -k "kernels_agree"
grep "^E  .*\|^FAILED\|passed\|failed" gpurun_out/r3_t26.log | cut -c1-300
