#!/bin/bash
source tools/gpu_call.sh
SETTLE_STEPS=240 step 300 r3_settle8.log python tools/dyn_settle_diag.py 16384
step 300 r3_replay.log python tools/replay_diag.py 32 0 8 16 24
