#!/bin/bash
source tools/gpu_call.sh
step 600 smoke24.log python -c "import __graft_entry__ as g; g.build(); g.smoke()"
tail -6 gpurun_out/smoke24.log
step 900 r3_t24.log python -m pytest tests -m gpu -x -q -p no:cacheprovider
tail -3 gpurun_out/r3_t24.log
step 300 bench24.json python bench.py --gpus 1 --steps 20 --warmup 5
grep -c '^{' gpurun_out/bench24.json
