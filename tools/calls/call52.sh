#!/bin/bash
source tools/gpu_call.sh
step 600 cur.log python -m pytest tests/test_hip_parity.py -x -q -k "config_variants or oracle_large"
tail -5 gpurun_out/cur.log
