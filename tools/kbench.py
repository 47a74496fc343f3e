#!/usr/bin/env python3
"""Developer tool: kernel times of the library named by $PARC_ENV_LIB (default: in-tree), bench.py's scenario
(untrained-policy actions, step + reset_done), hipEvents on the launch stream.  KB_DYN=0: kinematic step only."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(os.environ.get("KB_STEPS", "100"))
sys.stdout = sys.stderr
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
dyn = os.environ.get("KB_DYN", "1") == "1"
if int(os.environ.get("KB_MOTIONS", "0")) > 0:  # synthetic library (cfg 3 / 5)
    import tempfile
    from parc_amd.util import synth_dataset
    base = str(path_loader.resolve_path(cfg["env"]["dm"]["motion_file"]))
    cfg["env"]["dm"]["motion_file"] = synth_dataset.write_spec(os.path.join(tempfile.mkdtemp(), "motions.yaml"), base,
                                                               int(os.environ["KB_MOTIONS"]), yaw=os.environ.get("KB_YAW", "0") == "1")
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1234, mirror_ref_state=False, enable_dynamics=dyn)
env.reset()
lo, hi = env._action_bound_low, env._action_bound_high
torch.manual_seed(0)
mean, std = 0.5 * (hi + lo), 0.5 * (hi - lo)
ACT = [(mean + 0.05 * std * torch.randn(n, env._char_dof_pos.shape[1], device="cuda:0")).contiguous() for _ in range(4)]
for i in range(20):
    env.step(ACT[i & 3] if dyn else None); env.reset_done()
torch.cuda.synchronize()
env.set_kernel_timing(True)
t0 = time.perf_counter()
for i in range(steps):
    env.step(ACT[i & 3] if dyn else None); env.reset_done()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
kt = env.get_kernel_timing()
sys.stdout = sys.__stdout__
print(json.dumps({"lib": os.environ.get("PARC_ENV_LIB", "in-tree"), "envs": n, "step_ms": round(1e3 * dt / steps, 4),
                  "dyn_ms": round(kt["dynamics_ms"], 4), "obs_ms": round(kt["obs_ms"], 4),
                  "Menv_steps_s": round(n * steps / dt / 1e6, 2)}))
if os.environ.get("PARC_STAMPS"):  # -DPARC_STAMPS build: mean cycles per phase of k_env_post over the envs of the last step
    import ctypes as C
    arr = (C.c_double * 8)()
    env._lib.parc_env_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    rc = env._lib.parc_env_debug_stamps(env._handle, arr)
    names = ["prefetch+fill", "rays", "rows+contacts", "FK", "key obs", "reward+done", "obs stream", "-"]
    print(json.dumps({"rc": rc, "post_cycles": {k: round(v) for k, v in zip(names, arr)}, "total": round(sum(arr))}))
