#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "- -" "65536"
bash tools/vb.sh "-" "8192"
step 600 r3_t21.log python -m pytest tests -m gpu -q -p no:cacheprovider -k "dynamics or kernels or wave or flag"
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t21.log | cut -c1-300
