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
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1, mirror_ref_state=False)
env.reset()
for _ in range(5):
    env.step(None); env.reset_done()
torch.cuda.synchronize()
tot, post = env.profile_step(iters=30)
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
