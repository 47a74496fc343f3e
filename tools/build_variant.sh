# usage: [SCHED=<-amdgpu-sched-strategy value|none>] bash tools/build_variant.sh <out.so> [extra hipcc flags...]   (developer A/B builds; default = the flags of __graft_entry__.build())
out=$1; shift
S=${SCHED:-iterative-maxocc}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1048576"
if [ "$S" != "none" ]; then F="$F -mllvm -amdgpu-sched-strategy=$S"; fi
/opt/rocm/bin/hipcc $F "-DPARC_BUILD_FLAGS=\"$F $*\"" "$@" -o $out parc_amd/csrc/parc_env.hip
