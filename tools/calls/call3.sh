#!/bin/bash
source tools/gpu_call.sh
step 1100 r3_t3.log python -m pytest tests -m gpu -q -x --durations=8 -p no:cacheprovider
tail -12 gpurun_out/r3_t3.log
step 300 r3_replay3.log python tools/replay_diag.py 32
grep "steps_total" gpurun_out/r3_replay3.log | cut -c1-900
SETTLE_STEPS=240 step 300 r3_settle3.log python tools/dyn_settle_diag.py 16384
grep "t=4.0s\|t=8.0s" gpurun_out/r3_settle3.log
step 300 r3_bench3.json python bench.py --steps 200 --warmup 20 --no-cpu-baseline
step 200 r3_bench3_8192.json python bench.py --envs 8192 --steps 300 --no-cpu-baseline
python - <<'PY'
import json
for f in ["r3_bench3", "r3_bench3_8192"]:
    d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1]); r = d["roofline"]
    print(f, "%.2f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], r["kernel"][:20], "%.4f" % r["kernel_ms"], "obs %.4f" % r.get("obs_kernel", r)["kernel_ms"])
PY
