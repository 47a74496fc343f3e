"""Host-side scene assembly: YAML env config -> tables for the C-ABI (no GPU needed).

Reads the same keys ``IGParkourEnv.__init__`` / ``DeepMimicEnv.__init__`` / ``IGCharEnv`` read
(``ig_parkour_env.py:43-149``, ``dm_env.py:20-93``, ``ig_char_env.py:28-40,307-347``); keys the reference
requires (``env_config["..."]`` without default) are required here too and raise ``KeyError``; keys it
ignores on this path (``mgdm``, ``enable_replan_timer_obs``, ``plane``, ``camera_mode`` ...) are tolerated.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List

import numpy as np

from parc_amd import lib as L
from parc_amd import motion_lib, terrain
from parc_amd.char_model import CharModel, GeomType
from parc_amd.util import path_loader


@dataclass
class Scene:
    cfg: L.ParcEnvConfig
    char_model: CharModel
    clips: List[motion_lib.Clip]
    packed: dict
    grid: terrain.TerrainGrid
    ray_points: np.ndarray
    env_offsets: np.ndarray
    action_low: np.ndarray
    action_high: np.ndarray
    joint_err_w: np.ndarray
    dof_err_w: np.ndarray
    key_body_ids: List[int]
    obs_shapes: "dict"
    num_envs: int
    env_config: dict


def build_action_bounds_pd(cm: CharModel):
    """ig_char_env.py:307-347: spherical -> +-1.2 max|limit|; hinge -> mid +- 0.7 range."""
    dof_low, dof_high = cm.dof_limits()
    low = np.zeros(dof_high.shape)
    high = np.zeros(dof_high.shape)
    for j in range(1, cm.get_num_joints()):
        joint = cm.get_joint(j)
        d, idx = joint.get_dof_dim(), joint.dof_idx
        if d == 3:
            scale = 1.2 * max(np.max(np.abs(dof_low[idx:idx + 3])), np.max(np.abs(dof_high[idx:idx + 3])))
            low[idx:idx + 3] = -scale
            high[idx:idx + 3] = scale
        elif d == 1:
            mid = 0.5 * (dof_high[idx] + dof_low[idx])
            sc = 0.7 * (dof_high[idx] - dof_low[idx])
            low[idx] = mid - sc
            high[idx] = mid + sc
    return low, high


CONTROL_MODES = {"pd": 0, "vel": 1, "torque": 2, "pd_exp": 3, "pd_1d": 4}   # ig_char_env.py:21-26 = PARC_CTRL_* of include/parc_env.h


def build_action_bounds(cm: CharModel, control_mode: str):
    """ig_char_env.py:252-268: pd / pd_exp / pd_1d -> the pd bounds, vel -> +-2 pi (:349-353), torque -> +- the motor efforts (:355-362)."""
    if control_mode == "vel":
        n = cm.get_dof_size()
        return -2.0 * np.pi * np.ones([n]), 2.0 * np.pi * np.ones([n])
    if control_mode == "torque":
        eff = np.asarray(cm.dof_pd_params()[3], np.float64)
        return -eff, eff
    return build_action_bounds_pd(cm)


def env_offsets_square(num_envs, env_spacing, env_id_base=0, total_envs=None):
    """ig_parkour_env.py:386-398 for the global env ids [base, base + num_envs)."""
    total = num_envs if total_envs is None else total_envs
    per_row = max(int(np.sqrt(total)), 1)
    g = np.arange(env_id_base, env_id_base + num_envs)
    off = np.zeros((num_envs, 3), np.float32)
    off[:, 0] = (env_spacing * 2 * (g % per_row)).astype(np.float32)
    off[:, 1] = (env_spacing * 2 * (g // per_row)).astype(np.float32)
    return off


def parse_joint_err_weights(cm: CharModel, joint_err_w):
    """ig_parkour_env.py:1134-1152."""
    nj = cm.get_num_joints()
    w = np.ones(nj - 1, np.float32) if joint_err_w is None else np.array(joint_err_w, np.float32)
    assert w.shape[-1] == nj - 1
    dw = np.zeros(cm.get_dof_size(), np.float32)
    for j in range(1, nj):
        d = cm.get_joint_dof_dim(j)
        if d > 0:
            i = cm.get_joint_dof_idx(j)
            dw[i:i + d] = w[j - 1]
    return w, dw


def fill_dynamics(dp: L.ParcDynamicsParams, cm: CharModel, env_config: dict, sim_config: dict):
    geoms = [g for body in cm._geoms for g in body]
    if len(geoms) > L.MAX_GEOMS:
        raise ValueError("too many geoms")
    dp.num_geoms = len(geoms)
    for i, g in enumerate(geoms):
        dp.geom_body[i] = g.body_id
        dp.geom_type[i] = int(g.shape)
        size = np.zeros(3); size[: len(g.size)] = g.size
        for k in range(3):
            dp.geom_pos[i][k] = float(g.pos[k]); dp.geom_pos2[i][k] = float(g.pos2[k]); dp.geom_size[i][k] = float(size[k])
        dp.geom_density[i] = g.density
    kp, kd, arm, eff = cm.dof_pd_params()
    lo, hi = cm.dof_limits()
    for d in range(cm.get_dof_size()):
        dp.dof_stiffness[d] = kp[d]; dp.dof_damping[d] = kd[d]; dp.dof_armature[d] = arm[d]; dp.dof_effort[d] = eff[d]
        dp.dof_lower[d] = lo[d]; dp.dof_upper[d] = hi[d]
    sim_freq = env_config.get("sim_freq", 60)
    control_freq = env_config.get("control_freq", 10)
    assert sim_freq >= control_freq and sim_freq % control_freq == 0, \
        "Simulation frequency must be a multiple of the control frequency"  # ig_env.py:109-110
    physx = (sim_config or {}).get("physx", {})
    dp.gravity_z = env_config.get("gravity_z", -9.81)
    dp.sim_dt = 1.0 / sim_freq
    dp.sim_steps = int(sim_freq / control_freq)
    dp.substeps = int((sim_config or {}).get("substeps", 2))
    dp.solver_iterations = int(physx.get("num_position_iterations", 4))
    dp.friction = 1.0
    dp.restitution = 0.0
    dp.contact_offset = physx.get("contact_offset", 0.02)
    dp.max_depenetration_velocity = physx.get("max_depenetration_velocity", 10.0)
    dp.control_mode = CONTROL_MODES[env_config.get("control_mode", "pd")]
    dp.angular_damping = 0.01
    dp.max_angular_velocity = 100.0


# Developer / test switches of the library (ParcEnvConfig.dev_options).  The library itself reads no environment variable; the tools and
# tests may still say PARC_DEV_OPTIONS="key=value;..." or use the round-3 variable names, which are translated HERE, on the Python side.
_LEGACY_DEV_ENV = {"PARC_DYN_SEGMENTS": "segments", "PARC_DYN_DTANG": "dtang", "PARC_DYN_KERNEL": "kernel", "PARC_DYN_NO_RESIDUAL": "no_residual",
                   "PARC_EMA_LEADER": "ema_leader", "PARC_CURRICULUM_TWO_LAUNCHES": "curriculum_two_launches", "PARC_DYN_MANIFOLD_PERIOD": "man_period"}


def format_dev_options(dev_options=None):
    """dict / "k=v;k=v" string (+ the environment, see above) -> the bytes ParcEnvConfig.dev_options points at (None = no switch set)."""
    opts = {}
    for var, key in _LEGACY_DEV_ENV.items():
        if os.environ.get(var) is not None:
            opts[key] = os.environ[var]
    for src in (os.environ.get("PARC_DEV_OPTIONS"), dev_options):
        if isinstance(src, dict):
            opts.update({str(k): str(v) for k, v in src.items()})
        elif src:
            for kv in str(src).split(";"):
                if "=" in kv:
                    k, v = kv.split("=", 1)
                    opts[k.strip()] = v.strip()
    opts = {k: v for k, v in opts.items() if not (k == "kernel" and v == "wave")}   # "wave" = the default choice
    if not opts:
        return None
    return ";".join(f"{k}={v}" for k, v in sorted(opts.items())).encode()


def build_scene(config: dict, num_envs: int, device_index: int = 0, env_id_base: int = 0, total_envs=None,
                seed: int = 0, enable_dynamics=None, verbose=True, dev_options=None, env_offsets=None) -> Scene:
    env_config = config["env"]
    dm_config = env_config["dm"]

    # ---- required keys, exactly those the reference indexes without a default ------------------------
    fraction_dm_envs = env_config["fraction_dm_envs"]
    if min(int(fraction_dm_envs * num_envs), num_envs) != num_envs:
        raise ValueError("fraction_dm_envs < 1 is not supported: every env is a DeepMimic env")
    _ = env_config["contact_detection_eps"]  # read but unused by the reference too (ig_parkour_env.py:55,655)
    use_contact_info = bool(env_config["use_contact_info"])      # ig_parkour_env.py:72
    enable_tar_obs = bool(env_config.get("enable_tar_obs", True))  # :83
    control_mode = env_config.get("control_mode", "pd")                # ig_char_env.py:95 (ControlMode[...])
    if control_mode not in CONTROL_MODES:
        raise KeyError(control_mode)
    _ = env_config["debug_visuals"], env_config["ref_char_offset"], env_config["camera_mode"]

    char_file = str(path_loader.resolve_path(env_config["char_file"]))
    if not os.path.exists(char_file):  # e.g. "parc/data/assets/humanoid.xml" in the bundled civilization yaml
        alt = path_loader.REPO_ROOT / "data" / "assets" / os.path.basename(char_file)
        if alt.exists():
            char_file = str(alt)
    cm = CharModel(char_file)
    B, D = cm.get_num_bodies(), cm.get_dof_size()

    key_body_ids = [cm.get_body_id(n) for n in env_config.get("key_bodies", [])]
    tar_obs_steps = list(env_config.get("tar_obs_steps", [1]))
    ray = terrain.get_xy_points_cone(env_config["ray_dx"], env_config["ray_points_behind"], env_config["ray_points_ahead"],
                                     env_config["ray_num_left"], env_config["ray_num_right"], env_config["ray_angle"])
    env_off = env_offsets_square(num_envs, env_config["env_spacing"], env_id_base, total_envs)
    if env_offsets is not None:  # test hook: env origins given by the caller (the far-origin parity fixture) instead of the square layout
        env_off = np.ascontiguousarray(env_offsets, np.float32)
        assert env_off.shape == (num_envs, 3)
    if control_mode == "pd_1d":  # the reference's assert (ig_char_env.py:246-250)
        for j in range(1, cm.get_num_joints()):
            assert cm.get_joint_dof_dim(j) == 1, "pd_1d only supports 1D joints"
    act_low, act_high = build_action_bounds(cm, control_mode)
    jw, dw = parse_joint_err_weights(cm, env_config.get("joint_err_w", None))

    # ---- motions + terrain -------------------------------------------------------------------------
    clips = motion_lib.load_motion_file(dm_config["motion_file"], verbose=verbose)
    packed = motion_lib.pack_clips(clips, B)
    build_mode = dm_config.get("terrain_build_mode", "square")
    save_path = dm_config.get("terrain_save_path", None)
    if save_path is not None:
        save_path = str(path_loader.resolve_path(save_path))
    grid = None
    if save_path is not None and os.path.exists(save_path):  # ig_parkour_env.py:486-488
        grid = terrain.load_terrain(save_path)
    if grid is None:
        if any(c.terrain is None for c in clips):
            raise ValueError("every motion file needs terrain_data")
        subs = [terrain.SubTerrain.from_ms_terrain_data(c.terrain) for c in clips]
        if build_mode == "square":
            assert dm_config["terrains_per_motion"] == 1
            hm = dm_config["heightmap"]
            grid = terrain.build_terrain_square(subs, hm["horizontal_scale"], hm["padding"])
        elif build_mode == "wide":
            hm = dm_config["heightmap"]
            grid = terrain.build_terrain_wide(subs, hm["horizontal_scale"], hm["padding"], dm_config["terrains_per_motion"])
        elif build_mode == "file":
            grid = terrain.terrain_from_file(subs[0], num_envs)
            grid.motion_offsets = np.zeros((len(clips), 1, 2), np.float32)
        else:
            raise ValueError("unsupported terrain build mode")
        if save_path is not None:
            terrain.save_terrain(grid, save_path)

    # ---- C config ----------------------------------------------------------------------------------
    cfg = L.ParcEnvConfig()
    cfg.abi_version = L.ABI_VERSION
    cfg.struct_size = C.sizeof(L.ParcEnvConfig)
    cfg.device = device_index
    cfg.num_envs = num_envs
    cfg.model = L.make_char_model(cm)
    cfg.num_key_bodies = len(key_body_ids)
    for i, b in enumerate(key_body_ids):
        cfg.key_body_ids[i] = b
    if len(tar_obs_steps) > L.MAX_TAR_STEPS:
        raise ValueError("at most 6 tar_obs_steps")
    cfg.num_tar_obs_steps = len(tar_obs_steps)
    for i, s in enumerate(tar_obs_steps):
        cfg.tar_obs_steps[i] = int(s)
    cfg.num_rays = ray.shape[0]
    cfg.ray_points_host = L.np_f32p(ray)
    cfg.control_dt = 1.0 / env_config["control_freq"]
    cfg.episode_length = env_config["episode_length"]
    cfg.min_obs_h = env_config["min_obs_h"]
    cfg.max_obs_h = env_config["max_obs_h"]
    ws = [env_config[k] for k in ("pose_w", "vel_w", "root_pos_w", "root_vel_w", "key_pos_w")]
    tw = sum(ws)
    cfg.pose_w, cfg.vel_w, cfg.root_pos_w, cfg.root_vel_w, cfg.key_pos_w = [w / tw for w in ws]
    for j in range(B - 1):
        cfg.joint_err_w[j] = float(jw[j])
    for d in range(D):
        cfg.dof_err_w[d] = float(dw[d])
    cw = env_config["contact_weights"]
    ptd = env_config["pose_termination_dist"]
    assert len(cw) == cm.get_num_contact_bodies() == B and len(ptd) == B - 1
    for b in range(B):
        cfg.contact_weights[b] = float(cw[b])
    for j in range(B - 1):
        cfg.pose_termination_dist[j] = float(ptd[j])
    cfg.root_pos_termination_dist = env_config["root_pos_termination_dist"]
    cfg.root_rot_termination_angle = env_config["root_rot_termination_angle"]
    cfg.enable_early_termination = int(env_config["enable_early_termination"])
    cfg.pose_termination = int(env_config.get("pose_termination", False))
    cfg.track_root = int(env_config["track_root"])
    cfg.track_root_h = int(env_config["track_root_h"])
    cfg.report_tracking_error = int(env_config.get("report_tracking_error", False))
    cfg.fail_rate_ema_weight = 0.01  # dm_env.py:88
    cfg.min_motion_weight = dm_config.get("min_motion_weight", 0.01)
    cfg.rand_root_pos_offset_scale = env_config["rand_root_pos_offset_scale"]
    cfg.rand_reset = int(env_config["rand_reset"])
    cfg.demo_mode = int(env_config["demo_mode"])
    cfg.env_offsets_host = L.np_f32p(env_off)
    for d in range(D):
        cfg.action_low[d] = float(act_low[d])
        cfg.action_high[d] = float(act_high[d])
    dyn = env_config.get("hip", {}).get("enable_dynamics", True) if enable_dynamics is None else enable_dynamics  # the reference always simulates
    cfg.enable_dynamics = int(bool(dyn))
    cfg.body_pos_from_fk = int(env_config.get("hip", {}).get("body_pos_from_fk", True))
    fill_dynamics(cfg.dynamics, cm, env_config, config.get("sim", {}))
    cfg.seed = seed
    # contact_bodies (ig_parkour_env.py:62-63): the bodies that may touch the ground; non-empty switches on the fall termination
    names = cm.get_body_names()
    mask = 0
    for nm in env_config.get("contact_bodies", []) or []:
        if nm not in names:
            raise ValueError(f"contact_bodies: unknown body {nm!r}")
        mask |= 1 << names.index(nm)
    cfg.contact_body_mask = mask
    cfg.termination_height = float(env_config["termination_height"])
    cfg.global_obs = int(bool(env_config["global_obs"]))   # ig_parkour_env.py:83
    # _compute_obs passes root_height_obs=self._global_root_height_obs (ig_parkour_env.py:904): one more observation in front
    cfg.global_root_height_obs = int(bool(env_config["global_root_height_obs"]))
    cfg.use_contact_info = int(use_contact_info)
    cfg.enable_tar_obs = int(enable_tar_obs)
    cfg.dev_options = format_dev_options(dev_options)

    from collections import OrderedDict
    J, K, S, R = B - 1, len(key_body_ids), len(tar_obs_steps), ray.shape[0]
    obs_shapes = OrderedDict()  # ig_parkour_env.py:911-958
    obs_shapes["char_obs"] = {"use_normalizer": True, "shape": (int(bool(env_config["global_root_height_obs"])) + 6 + 3 + 3 + 6 * J + D + 3 * K,)}
    if enable_tar_obs:
        obs_shapes["tar_obs"] = {"use_normalizer": True, "shape": (S, 3 + 6 + 6 * J + 3 * K)}
    if use_contact_info:
        if enable_tar_obs:
            obs_shapes["tar_contacts"] = {"use_normalizer": False, "shape": (S, B)}
        obs_shapes["char_contacts"] = {"use_normalizer": False, "shape": (B,)}
    obs_shapes["hf"] = {"use_normalizer": False, "shape": (R,)}

    return Scene(cfg=cfg, char_model=cm, clips=clips, packed=packed, grid=grid, ray_points=ray, env_offsets=env_off,
                 action_low=act_low, action_high=act_high, joint_err_w=jw, dof_err_w=dw, key_body_ids=key_body_ids,
                 obs_shapes=obs_shapes, num_envs=num_envs, env_config=env_config)
