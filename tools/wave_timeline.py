#!/usr/bin/env python3
"""Developer tool: hand-off timeline of one substep of k_dynamics_wave (block 0, third substep) from a -DPARC_TIMELINE build
(PARC_ENV_LIB=<variant>): absolute cycle-counter values per wave and event, printed relative to the earliest one."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1, mirror_ref_state=False, enable_dynamics=True)
env.reset()
lo, hi = env._action_bound_low, env._action_bound_high
torch.manual_seed(0)
ACT = (0.5 * (hi + lo) + 0.05 * 0.5 * (hi - lo) * torch.randn(n, env._char_dof_pos.shape[1], device="cuda:0")).contiguous()
W0 = {0: "substep start", 1: "arm FK done", 2: "arm inward done", 9: "head start", 11: "head done", 6: "torso start", 7: "torso record in", 8: "torso children in",
      3: "pelvis start", 4: "pelvis record in", 5: "pelvis children in", 12: "ACC published", 13: "trunk outward done", 14: "next trunk kin published", 15: "own limb outward done"}
WL = {0: "parent kin in", 1: "FK done", 2: "inward done / UP", 5: "REC head out", 4: "REC torso out", 3: "REC pelvis out", 6: "ACC in", 15: "outward done"}
env._lib.parc_env_debug_wave_timeline.argtypes = [C.POINTER(C.c_double)]
for rep in range(3):
    for _ in range(10):
        env.step(ACT); env.reset_done()
    a = (C.c_double * 64)()
    env._lib.parc_env_debug_wave_timeline(a)
    t0 = min(v for v in a if v > 0)
    ev = []
    for w in range(4):
        names = W0 if w == 0 else WL
        for i, nm in names.items():
            v = a[w * 16 + i]
            if v > 0:
                ev.append((v - t0, w, nm))
    print("--- sample", rep)
    for t, w, nm in sorted(ev):
        print("%8d  w%d  %s" % (t, w, nm))
