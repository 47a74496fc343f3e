#!/bin/bash
source tools/gpu_call.sh
PARC_DYN_SEGMENTS=none PARC_DYN_DTANG=1e4 step 200 kb_none_1e4.json python tools/kbench.py 65536
PARC_DYN_SEGMENTS=none step 200 kb_none.json python tools/kbench.py 65536
PARC_DYN_SEGMENTS=capsules step 200 kb_caps.json python tools/kbench.py 65536
step 200 kb_all.json python tools/kbench.py 65536
PARC_DYN_SEGMENTS=none step 200 kb_none_8192.json python tools/kbench.py 8192
step 200 kb_all_8192.json python tools/kbench.py 8192
cat gpurun_out/kb_*.json
step 900 r3_t4.log python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider -k "dynamics or kernels or wave"
tail -30 gpurun_out/r3_t4.log | cut -c1-300
