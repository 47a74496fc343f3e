#!/bin/bash
source tools/gpu_call.sh
step 1100 prof_r03.log bash tools/profile_round.sh r03
tail -5 gpurun_out/prof_r03.log
