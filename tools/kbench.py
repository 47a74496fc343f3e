#!/usr/bin/env python3
"""Developer tool: time the step kernel of the library named by $PARC_ENV_LIB (default: in-tree) with hipEvents."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sys.stdout = sys.stderr
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
dyn = os.environ.get("KB_DYN", "0") == "1"
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1, mirror_ref_state=False, enable_dynamics=dyn)
env.reset()
if dyn:
    lo, hi = env._action_bound_low, env._action_bound_high
    torch.manual_seed(0)
    ACT = (0.5 * (hi + lo) + 0.025 * (hi - lo) * torch.randn(n, env._char_dof_pos.shape[1], device="cuda:0")).contiguous()
for _ in range(5):
    env.step(ACT if dyn else None); env.reset_done()
torch.cuda.synchronize()
tot, post = env.profile_step(iters=30, action=ACT if dyn else None)
sys.stdout = sys.__stdout__
print(json.dumps({"lib": os.environ.get("PARC_ENV_LIB", "in-tree"), "envs": n, "step_ms": tot, "post_ms": post,
                  "GBps": 5772 * n / (post * 1e-3) / 1e9}))
if os.environ.get("PARC_STAMPS"):
    import ctypes as C
    arr = (C.c_double * 8)()
    env._lib.parc_env_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    rc = env._lib.parc_env_debug_stamps(env._handle, arr)
    names = ["prefetch+fill", "rays", "rows+contacts", "FK", "key obs", "reward+done", "obs stream", "-"]
    tot_c = sum(arr)
    print(json.dumps({"rc": rc, "cycles": {n: round(v) for n, v in zip(names, arr)}, "total": round(tot_c)}))
    if dyn and hasattr(env._lib, "parc_env_debug_dyn_stamps"):
        a16 = (C.c_double * 16)()
        env._lib.parc_env_debug_dyn_stamps.argtypes = [C.POINTER(C.c_double)]
        env._lib.parc_env_debug_dyn_stamps(a16)
        nm = ["load", "fk", "loop-head", "own-inertia", "contacts", "children", "root-solve", "joint", "inward-tail", "outward", "integrate"]
        tot = sum(a16)
        print(json.dumps({"dyn_phase_share": {k: round(v / tot, 4) for k, v in zip(nm, a16)}}))
