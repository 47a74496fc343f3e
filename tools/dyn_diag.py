"""Dev tool: open-loop rollout statistics of the dynamics kernel (run once per PARC_DYN_KERNEL setting)."""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from gpu_helpers import default_config, write_motion_yaml, to_np
from parc_amd.envs.hip_parkour_env import HipParkourEnv

n = 4096
cfg = default_config()
tmp = pathlib.Path(tempfile.mkdtemp())
cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp, ["civilization"], [1.0])
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=9, enable_dynamics=True, mirror_ref_state=True)
env.reset()
fmax = []
for it in range(60):
    act = env._ref_dof_pos.clone()
    obs, rew, done, info = env.step(act)
    f = np.abs(to_np(env._char_contact_forces)).reshape(n, -1).max(1)
    fmax.append(f.max())
    if it % 10 == 9:
        print(os.environ.get("PARC_DYN_KERNEL", "coop"), it, "fmax %.0f q999 %.0f q99 %.0f  done %.3f rew %.3f finite %s" % (
            f.max(), np.quantile(f, 0.999), np.quantile(f, 0.99), to_np(done).astype(bool).mean(), to_np(rew).mean(),
            bool(torch.isfinite(obs).all())), flush=True)
    env.reset_done()
print("max over run", max(fmax))
