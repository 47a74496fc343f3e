#!/bin/bash
# Developer tool (run on the GPU box via gpurun): PMC passes for the step kernel. Counters go to gpurun_out/pmc_*.
# Usage: tools/pmc.sh "<counter list>" <tag> [kernel-name substring]   (KB_DYN=1 runs the dynamics step)
set -e
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$2 -- python3 $ROOT/tools/kbench.py 65536 > /dev/null 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$2/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"][:40]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    if "${3:-k_env_post}" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
