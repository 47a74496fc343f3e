#!/bin/bash
source tools/gpu_call.sh
step 1100 gpu_all.log python -m pytest tests -x -q -m gpu
tail -3 gpurun_out/gpu_all.log
step 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -2 gpurun_out/smoke.log
step 600 bench_driver.json python bench.py --gpus 1 --steps 20 --warmup 5
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_driver.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['obs_kernel']['frac'], d['cpu_baseline']['value'], d['dynamics_timeouts'])
PY
