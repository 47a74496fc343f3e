#!/bin/bash
source tools/gpu_call.sh
step 200 cmp_wave_coop.log python tools/dyn_cmp.py wave coop 512
step 200 cmp_wave_thread.log python tools/dyn_cmp.py wave thread 512
PARC_DYN_SEGMENTS=none step 200 cmp_wave_coop_noseg.log python tools/dyn_cmp.py wave coop 512
grep -v Loading gpurun_out/cmp_wave_coop.log | cut -c1-220
echo ---- thread; grep -v Loading gpurun_out/cmp_wave_thread.log | cut -c1-220
echo ---- noseg; grep -v Loading gpurun_out/cmp_wave_coop_noseg.log | cut -c1-220
