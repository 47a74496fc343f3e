#!/bin/bash
# Run on the GPU box: the benchmark configurations quoted in DESIGN.md section 9 -> gpurun_out/bench_set_<tag>/*.json
TAG=${1:-r02}
OUT=gpurun_out/bench_set_$TAG
mkdir -p $OUT
CIV=data/configs/tracker_config/dm_env_civilization.yaml
run() { name=$1; shift; python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || tail -3 $OUT/$name.err; python - <<PY
import json
d = json.load(open("$OUT/$name.json"))
r = d["roofline"]
print("%-22s %8.2f M env-steps/s  %.4f ms/step  kernel %s %.4f ms" % ("$name", d["value"] / 1e6, d["ms_per_step"], r["kernel"][:24], r["kernel_ms"]))
PY
}
run headline --steps 200 --warmup 20
run kinematic_65536 --dynamics 0 --steps 200 --no-cpu-baseline
run cfg2 --envs 4096 --dynamics 0 --config $CIV --steps 300
run cfg2_graph --envs 4096 --dynamics 0 --config $CIV --steps 300 --graph 1 --no-cpu-baseline
run cfg3 --envs 16384 --motions 1024 --steps 200 --no-cpu-baseline
run cfg5_shard --envs 16384 --motions 16384 --yaw 1 --steps 200 --no-cpu-baseline
run shard_8192 --envs 8192 --steps 300 --no-cpu-baseline
run shard_8192_graph --envs 8192 --steps 300 --graph 1 --no-cpu-baseline
PARC_BENCH_SHARE_GPU=1 run two_ranks_one_gpu --gpus 2 --steps 100 --no-cpu-baseline
PARC_BENCH_SHARE_GPU=1 run two_ranks_one_gpu_weak --gpus 2 --scaling weak --envs 32768 --steps 100 --no-cpu-baseline
