#!/usr/bin/env python3
"""Developer tool: how steady are the reported contact flags (|F_b| > 1e-5, what the observation and the recorder use) of characters at rest?
Hold the reset pose for SETTLE steps on the cfg-3 style scene, then watch the flags of the envs that are at rest for 30 steps: the share of
(env, body) pairs whose flag toggles although the env does not move, and the per-body on-rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=7, mirror_ref_state=False, enable_dynamics=True)
env.reset()
hold = env._char_dof_pos.clone()
for _ in range(int(os.environ.get("SETTLE", "240"))):
    env.step(hold)
flags = []; speed = []
for _ in range(30):
    env.step(hold)
    flags.append((env._char_contact_forces.norm(dim=-1) > 1e-5).cpu().numpy())
    speed.append(env._char_root_vel.norm(dim=-1).cpu().numpy())
flags = np.stack(flags); speed = np.stack(speed)
rest = speed.max(0) < 0.02                       # envs that do not move during the window
f = flags[:, rest]                               # [30, envs at rest, 15]
on = f.mean(0)                                   # on-rate per (env, body)
steady = (on == 0) | (on == 1)
touch = on > 0
print("envs at rest %d of %d" % (rest.sum(), n))
print("(env, body) pairs that touch at all: %.3f of all; of those, flag on in EVERY step: %.3f, mean on-rate %.3f" % (touch.mean(), (on[touch] == 1).mean(), on[touch].mean()))
names = env._kin_char_model.get_body_names()
for b in range(15):
    t = touch[:, b]
    if t.sum() > 20: print("  %-16s touching envs %5d  always-on %.3f  mean on-rate %.3f" % (names[b], t.sum(), (on[t, b] == 1).mean(), on[t, b].mean()))
