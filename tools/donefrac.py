import sys, torch
sys.path.insert(0, '.')
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
env = HipParkourEnv(cfg, 65536, "cuda:0", False, seed=1, mirror_ref_state=False)
env.reset()
fr = []
for i in range(40):
    _, r, d, _ = env.step(None)
    fr.append(((d != 0).float().mean().item(), (d == 1).float().mean().item(), r.mean().item()))
    env.reset_done()
print([tuple(round(x, 3) for x in f) for f in fr])
print(torch.bincount(env._motion_ids.long(), minlength=5))
