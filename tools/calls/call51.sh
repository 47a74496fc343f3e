#!/bin/bash
source tools/gpu_call.sh
mkdir -p gpurun_out/bench_set_r03
step 600 bench_set_r03/headline.json python bench.py --steps 200 --warmup 20
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_set_r03/headline.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['cpu_baseline']['cpu_model'], d['cpu_baseline']['value'])
PY
