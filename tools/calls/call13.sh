#!/bin/bash
source tools/gpu_call.sh
PARC_ENV_LIB=variants/libparc_env_stamps.so step 300 stamps13.log python tools/wave_stamps.py 65536
python - <<'PY'
import json
t = open("gpurun_out/stamps13.log").read()
d = json.loads(t[t.index("{"):t.rindex("}") + 1])["cycles_per_control_step"]
keys = list(d["wave0"].keys())
print("%-18s" % "", *["%9s" % w for w in d])
for k in keys:
    print("%-18s" % k, *["%9d" % d[w][k] for w in d])
PY
