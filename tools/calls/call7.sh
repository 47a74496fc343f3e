#!/bin/bash
source tools/gpu_call.sh
DYN_CMP_DUMP=gpurun_out/cmp_state.npz step 200 cmp_wave_thread2.log python tools/dyn_cmp.py wave thread 128
