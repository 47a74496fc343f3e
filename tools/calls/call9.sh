#!/bin/bash
source tools/gpu_call.sh
step 300 nan_hunt.log python tools/nan_hunt.py 16384 1024
grep -v "Loading\|Synthetic" gpurun_out/nan_hunt.log | cut -c1-300
