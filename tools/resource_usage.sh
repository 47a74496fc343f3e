#!/bin/bash
# Register / scratch / LDS usage of the library's main kernels (hipcc -Rpass-analysis=kernel-resource-usage); no GPU needed.
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -pragma-unroll-threshold=1048576"
/opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -o /tmp/parc_ru.so "$(dirname "$0")/../parc_amd/csrc/parc_env.hip" 2>&1 | \
  python3 -c "
import re, sys
cur = None
for ln in sys.stdin:
    m = re.search(r'remark: Function Name: (\S+)', ln)
    if m: cur = m.group(1); vals = {}; continue
    m = re.search(r'remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)', ln)
    if m and cur:
        vals[m.group(1)] = int(m.group(2))
        if m.group(1).startswith('LDS'):
            if any(k in cur for k in ('k_dynamics_wave', 'k_env_post', 'k_dynamics_coop', 'k_tail', 'k_fail')):
                print('%-70s' % cur[:70], ' '.join('%s=%d' % (k.split(' ')[0], v) for k, v in vals.items()))
    if 'error' in ln: print(ln, end='')
"
