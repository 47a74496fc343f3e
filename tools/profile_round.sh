#!/bin/bash
# Run on the GPU box (via gpurun): everything the committed profiles/ summaries of a round come from.
#   tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/...
# (1) rocprofv3 --kernel-trace --stats of the default bench.py run; (2) separate --pmc passes (the MI355X guide: counters
# in their own runs, --kernel-trace only) for the HBM traffic of every kernel (bench.py) and the SQ counters of the
# dynamics kernel; (3) the same traffic passes on the cfg-5 shard (16 384 envs x 16 384-clip library).
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 100 --warmup 10 > $OUT/bench_under_profiler.json 2> $OUT/stats.log
echo "stats pass done"
pmc() { # <dir> <counters> <bench args...>
  local d=$1 c=$2; shift 2
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$d -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-sample-steps 0 "$@" > /dev/null 2> $OUT/$d.log
  echo "$d done"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
pmc sq2 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pmc sq3 "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32"
pmc sq4 "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
pmc c5fetch FETCH_SIZE --envs 16384 --motions 16384 --yaw 1
pmc c5write WRITE_SIZE --envs 16384 --motions 16384 --yaw 1
cd $ROOT
python3 tools/profile_summary.py $OUT $TAG
