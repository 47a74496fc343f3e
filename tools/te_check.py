"""Dev check: test roll-out with report_tracking_error on (kernel output -> TrackingErrorTracker -> test info keys)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.learning.dm_ppo_agent import DMPPOAgent
from parc_amd.util import path_loader
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
cfg["env"]["report_tracking_error"] = True
acfg = path_loader.load_config("data/configs/tracker_config/dm_agent_default.yaml")
env = HipParkourEnv(cfg, 512, "cuda:0", False, seed=3)
agent = DMPPOAgent(acfg, env, "cuda:0")
res = agent.test_model(num_episodes=1024)
for k, v in res.items():
    if "test_mean" in k or k in ("mean_return", "mean_ep_len", "num_eps"):
        print(k, v)
assert all(res[k] > 0 for k in res if "test_mean" in k) and len([k for k in res if "test_mean" in k]) == 7
print("ok")
