#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel-trace stats of tools/kbench.py.  Usage: tools/kstats.sh <tag> <envs>
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
KB_STEPS=100 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kstats_$1 -- python3 $ROOT/tools/kbench.py $2 > $ROOT/gpurun_out/kstats_$1.json 2>$ROOT/gpurun_out/kstats_$1.log
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kstats_$1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-60s calls %6s avg %9.1f us  total %6.2f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
