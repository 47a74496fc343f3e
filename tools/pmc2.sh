#!/bin/bash
# Developer tool (GPU box): PMC passes over tools/kbench.py for one kernel.  Usage: tools/pmc2.sh <tag> <kernel substring> <envs> "<counters pass 1>" ["<counters pass 2>" ...]
set -e
TAG=$1; KSUB=$2; ENVS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  KB_STEPS=20 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$i -- python3 $ROOT/tools/kbench.py $ENVS > /dev/null 2>$ROOT/gpurun_out/pmc_${TAG}_$i.log || tail -3 $ROOT/gpurun_out/pmc_${TAG}_$i.log
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "$KSUB" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
out = {c: acc[c] / cnt[c] for c in acc}
print(json.dumps({"kernel": "$KSUB", "envs": $ENVS, "avg_per_dispatch": out}, indent=1))
json.dump({"kernel": "$KSUB", "envs": $ENVS, "avg_per_dispatch": out}, open("gpurun_out/pmc_${TAG}.json", "w"), indent=1)
PY
