#!/bin/bash
source tools/gpu_call.sh
step 600 smoke25.log python -c "import __graft_entry__ as g; g.build(); g.smoke()"
grep "smoke ok\|build ok" gpurun_out/smoke25.log; tail -4 gpurun_out/smoke25.log.err
