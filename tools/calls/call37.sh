#!/bin/bash
source tools/gpu_call.sh
step 1100 gpu_all.log python -m pytest tests -x -q -m gpu
tail -5 gpurun_out/gpu_all.log
step 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -3 gpurun_out/smoke.log
