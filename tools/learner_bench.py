"""Dev tool: PPO iteration time with and without the bf16 trunk option (learner side only; not the headline metric)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.learning.dm_ppo_agent import DMPPOAgent
from parc_amd.util import path_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for amp in ("none", "bf16"):
    cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
    acfg = path_loader.load_config("data/configs/tracker_config/dm_agent_default.yaml")
    acfg["model"]["amp"] = amp
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=2, enable_dynamics=True, mirror_ref_state=False)
    agent = DMPPOAgent(acfg, env, "cuda:0")
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    agent._train_iter(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        info = agent._train_iter()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"amp={amp}: {dt * 1e3:.1f} ms per iteration, {32 * n / dt / 1e3:.0f}k env-steps/s, loss {info['loss'].item():.4f}", flush=True)
    del agent, env
