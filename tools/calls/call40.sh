#!/bin/bash
source tools/gpu_call.sh
step 120 latency.log ./variants/latency
cat gpurun_out/latency.log
