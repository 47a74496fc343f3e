#!/bin/bash
source tools/gpu_call.sh
bash tools/vb.sh "- variants/libparc_env_s_none.so variants/libparc_env_s_gcn-max-ilp.so variants/libparc_env_s_gcn-max-memory-clause.so variants/libparc_env_s_iterative-ilp.so variants/libparc_env_s_iterative-minreg.so -" "65536"
PARC_DYN_SEGMENTS=none step 100 kb19_none.json python tools/kbench.py 65536
PARC_DYN_SEGMENTS=capsules step 100 kb19_caps.json python tools/kbench.py 65536
cat gpurun_out/kb19_none.json gpurun_out/kb19_caps.json
PARC_ENV_LIB=variants/libparc_env_stamps.so step 300 stamps19.log python tools/wave_stamps.py 65536
python - <<'PY'
import json
t = open("gpurun_out/stamps19.log").read()
d = json.loads(t[t.index("{"):t.rindex("}") + 1])["cycles_per_control_step"]
keys = list(d["wave0"].keys())
print("%-18s" % "", *["%9s" % w for w in d])
for k in keys:
    print("%-18s" % k, *["%9d" % d[w][k] for w in d])
PY
PARC_ENV_LIB=variants/libparc_env_tl.so step 300 tl19.log python tools/wave_timeline.py 65536
grep -A40 "sample 2" gpurun_out/tl19.log
