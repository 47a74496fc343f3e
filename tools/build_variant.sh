# usage: bash tools/build_variant.sh <out.so> [extra hipcc flags...]   (developer A/B builds; same flags as __graft_entry__.build())
out=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -pragma-unroll-threshold=1048576"
/opt/rocm/bin/hipcc $F "-DPARC_BUILD_FLAGS=\"$F $*\"" "$@" -o $out parc_amd/csrc/parc_env.hip
