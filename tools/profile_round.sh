#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench + HBM-traffic PMC passes (separate runs, as the
# MI355X guide prescribes).  Usage: tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/{stats,fetch,write}/..., summary json
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 100 --warmup 10 > $OUT/bench_under_profiler.json 2> $OUT/stats.log
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/fetch.log
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/write.log
echo "write pass done"
cd $ROOT
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
def pmc(d, name):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != name: continue
        k = row["Kernel_Name"].split("(")[0]
        acc[k] += float(row["Counter_Value"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}
fe, wr = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 20 (default config: dynamics on, 65536 envs). "
                   "Units: KiB per dispatch; gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md HBM section): consumers double it.",
           "FETCH_SIZE_KiB_avg_per_dispatch": fe, "WRITE_SIZE_KiB_avg_per_dispatch": wr}, open(out + "/pmc_hbm_traffic.json", "w"), indent=1)
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
print("stats file:", st)
for k in fe: print("%-40s fetch %10.1f KiB  write %10.1f KiB" % (k[:40], fe[k], wr.get(k, 0)))
PY
