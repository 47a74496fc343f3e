#!/usr/bin/env python3
"""Developer tool: manifold-size statistics of k_dynamics_wave from a -DPARC_STAMPS -DPARC_COUNTS build (PARC_ENV_LIB=variants/libparc_env_counts.so):
per body, the histogram over waves of the largest per-lane number of contacts within the speculative margin at substep 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1234, mirror_ref_state=False, enable_dynamics=True)
env.reset()
lo, hi = env._action_bound_low, env._action_bound_high
torch.manual_seed(0)
ACT = (0.5 * (hi + lo) + 0.05 * 0.5 * (hi - lo) * torch.randn(n, env._char_dof_pos.shape[1], device="cuda:0")).contiguous()
hold = os.environ.get("MH_HOLD", "0") == "1"   # hold the reset pose (settle scenario: characters end up lying on the terrain)
for _ in range(int(os.environ.get("MH_WARM", "20"))):
    env.step(env._char_dof_pos.clone() if hold else ACT)
    if not hold: env.reset_done()
torch.cuda.synchronize()
a = (C.c_double * 1024)()
env._lib.parc_env_debug_wave_hist.argtypes = [C.POINTER(C.c_double)]
env._lib.parc_env_debug_wave_hist(a)
for _ in range(10):
    env.step(env._char_dof_pos.clone() if hold else ACT)
    if not hold: env.reset_done()
torch.cuda.synchronize()
env._lib.parc_env_debug_wave_hist(a)
def row(b, k): return [int(a[(b * 4 + k) * 16 + i]) for i in range(16)]
for w in range(4):
    r = row(15, w); t = max(sum(r), 1)
    print("wave %d: planes held after a discovery, largest lane of the wave (%% of waves, 0..15+): " % w + " ".join("%5.1f" % (100.0 * x / t) for x in r))
print("drops", env._lib.parc_env_dynamics_manifold_drops(env._handle))
