#!/bin/bash
source tools/gpu_call.sh
step 200 cmp8_wave_coop.log python tools/dyn_cmp.py wave coop 512
step 200 cmp8_wave_thread.log python tools/dyn_cmp.py wave thread 512
grep -v Loading gpurun_out/cmp8_wave_coop.log | cut -c1-200 | head -12
echo ---- thread; grep -v Loading gpurun_out/cmp8_wave_thread.log | cut -c1-200 | head -12
PARC_DYN_SEGMENTS=none step 200 kb8_none.json python tools/kbench.py 65536
step 200 kb8_all.json python tools/kbench.py 65536
step 200 kb8_all_8192.json python tools/kbench.py 8192
cat gpurun_out/kb8_*.json
step 900 r3_t8.log python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider -k "dynamics or kernels or wave"
grep "^E  .*Error\|^FAILED\|passed\|failed" gpurun_out/r3_t8.log | cut -c1-400
