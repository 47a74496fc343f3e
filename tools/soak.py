import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader
n = 65536
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=99, mirror_ref_state=False, enable_dynamics=True)
env.reset()
lo, hi = env._action_bound_low, env._action_bound_high
g = torch.Generator(device="cuda:0"); g.manual_seed(3)
mean, std = 0.5 * (hi + lo), 0.5 * (hi - lo)
t0 = time.time(); tot_done = 0
for it in range(4000):
    if it % 4 == 0:
        act = (mean + 0.3 * std * torch.randn(n, env._char_dof_pos.shape[1], device="cuda:0", generator=g)).contiguous()
    obs, rew, done, info = env.step(act)
    if it % 500 == 499:
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(env._char_root_pos).all() and torch.isfinite(env._char_dof_vel).all())
        to = env._lib.parc_env_dynamics_timeouts(env._handle)
        print(it + 1, "finite", ok, "timeouts", to, "done frac", float((done != 0).float().mean()), "max |dofvel|", float(env._char_dof_vel.abs().max()), flush=True)
        assert ok and to == 0
    env.reset_done()
print("soak ok", time.time() - t0)
