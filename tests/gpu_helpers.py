"""Builders for the -m gpu tests: the golden 4-clip scene and larger synthetic scenes on the HIP env."""
import copy
import os

import numpy as np
import yaml

from conftest import DATA, golden
from parc_amd.util import path_loader

GOLDEN_WEIGHTS = [1.0, 1.5, 2.0, 2.5]


def default_config():
    """dm_env_default.yaml with the physics switched off: the parity tests inject the character state (the kinematic
    configuration); tests of the dynamics pass ``enable_dynamics=True`` to the env."""
    cfg = copy.deepcopy(path_loader.load_config(os.path.join(DATA, "configs/tracker_config/dm_env_default.yaml")))
    cfg["env"].setdefault("hip", {})["enable_dynamics"] = False
    return cfg


def write_motion_yaml(tmp_dir, clip_names, weights):
    p = os.path.join(str(tmp_dir), "motions.yaml")
    with open(p, "w") as f:
        yaml.safe_dump({"motions": [{"file": os.path.join(DATA, "motion_terrains", c + ".pkl"), "weight": float(w)}
                                    for c, w in zip(clip_names, weights)]}, f)
    return p


def golden_env(tmp_dir, body_pos_from_fk=False, tracking=True, num_envs=64):
    """HipParkourEnv on the scene of tests/golden/env_step.npz."""
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g = golden("env_step")
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_dir, [str(c) for c in g["clips"]], GOLDEN_WEIGHTS)
    cfg["env"]["report_tracking_error"] = tracking
    cfg["env"]["hip"]["body_pos_from_fk"] = body_pos_from_fk
    env = HipParkourEnv(cfg, num_envs, "cuda:0", False)
    return env, g


_IN = {"char_root_pos": "_char_root_pos", "char_root_rot": "_char_root_rot", "char_root_vel": "_char_root_vel",
       "char_root_ang_vel": "_char_root_ang_vel", "char_dof_pos": "_char_dof_pos", "char_dof_vel": "_char_dof_vel",
       "char_body_pos": "_char_rigid_body_pos", "contact_forces": "_char_contact_forces", "motion_ids": "_motion_ids",
       "terrain_ids": "_motion_terrain_ids", "time_offsets": "_motion_time_offsets", "timestep": "_timestep_buf",
       "time": "_time_buf", "done": "_done_buf"}


def inject(env, g, prefix):
    import torch
    for gk, attr in _IN.items():
        t = getattr(env, attr)
        t.copy_(torch.from_numpy(np.ascontiguousarray(g[prefix + gk])).to(t.dtype).to(t.device))
    env.set_fail_rates(g[prefix + "fail_rates"])


def to_np(t):
    return t.detach().cpu().numpy()
