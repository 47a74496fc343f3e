#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

How the reference is driven:
  * a ``parc`` package alias whose ``__path__`` is ``/root/reference/PARC`` (the scripts
    import lowercase ``parc``), empty module stubs for ``trimesh / wandb / gym /
    isaacgym*`` (none is needed for the arithmetic), ``sys.dont_write_bytecode``;
  * the reference's ``file_io.load_ms_file`` (which unpickles) is replaced by our
    data-only decoder (``parc_amd.ms_file``) so that no bundled ``.pkl`` is unpickled;
  * ``IGParkourEnv`` / ``DeepMimicEnv`` cannot be constructed (Isaac Gym), so bare
    instances are made with ``object.__new__`` and given exactly the attributes their
    methods read; then the reference's OWN methods are called in the reference's order:
    ``IGEnv._post_physics_step`` = refresh hf rays -> ``_update_time`` ->
    ``_update_ref_motion`` -> ``_compute_obs`` -> ``_update_reward`` -> ``_update_done``
    (``ig_env.py:368-377``), and ``DeepMimicEnv.reset`` for the reset fixture.

Outputs are data only (inputs + expected outputs as ``.npz``), plus re-encoded copies
of the five bundled clips under ``data/motion_terrains`` written by OUR writer, and the
MJCF asset.  No reference source text is stored.
"""
import os
import shutil
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

for _name in ["trimesh", "wandb", "gym", "gym.spaces", "isaacgym", "isaacgym.gymapi",
              "isaacgym.gymtorch", "isaacgym.gymutil"]:
    sys.modules[_name] = types.ModuleType(_name)
_parc = types.ModuleType("parc")
_parc.__path__ = [os.path.join(REF, "PARC")]
sys.modules["parc"] = _parc

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import parc.util.file_io as ref_file_io  # noqa: E402
from parc_amd import ms_file  # noqa: E402


def _safe_load(filepath):
    d = ms_file.load_ms_file(filepath, load_misc=False)
    md = None if d.motion_data is None else ref_file_io.MSMotionData(**vars(d.motion_data))
    td = None if d.terrain_data is None else ref_file_io.MSTerrainData(**vars(d.terrain_data))
    return ref_file_io.MSFileData(motion_data=md, terrain_data=td, misc_data=None)


ref_file_io.load_ms_file = _safe_load

import parc.anim.kin_char_model as kcm  # noqa: E402
import parc.anim.motion_lib as ref_mlib  # noqa: E402
import parc.motion_tracker.envs.base_env as base_env  # noqa: E402
import parc.motion_tracker.envs.ig_char_env as ig_char_env  # noqa: E402
import parc.motion_tracker.envs.ig_env as ig_env  # noqa: E402
import parc.motion_tracker.envs.ig_parkour.dm_env as dm_env  # noqa: E402
import parc.motion_tracker.envs.ig_parkour.ig_parkour_env as ipe  # noqa: E402
import parc.motion_tracker.envs.ig_parkour.mgdm_dm_util as mdu  # noqa: E402
import parc.util.geom_util as geom_util  # noqa: E402
import parc.util.terrain_util as terrain_util  # noqa: E402
import parc.util.torch_util as tu  # noqa: E402

torch.set_num_threads(1)
DEV = "cpu"
CLIPS = ["sfu", "civilization", "TEASER_TERRAIN", "dec2024_teaser_717_1_opt_dm"]


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().copy()
    return np.asarray(x)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print("wrote", path, {k: tuple(npy(v).shape) for k, v in arrs.items()})


# ----------------------------------------------------------------------------------
def export_data_files():
    """Re-encode the bundled clips with OUR writer + copy the MJCF data asset."""
    os.makedirs(os.path.join(REPO, "data", "motion_terrains"), exist_ok=True)
    os.makedirs(os.path.join(REPO, "data", "assets"), exist_ok=True)
    shutil.copyfile(os.path.join(REF, "data/assets/humanoid.xml"), os.path.join(REPO, "data/assets/humanoid.xml"))
    for clip in CLIPS + ["dec2024_teaser_717_1_modified_opt"]:
        d = ms_file.load_ms_file(os.path.join(REF, "data/motion_terrains", clip + ".pkl"))
        misc = None
        if d.misc_data is not None and "obs" in d.misc_data:
            # Isaac-Gym-recorded observation stream: keep as its own fixture
            shapes = d.misc_data["obs_shapes"]
            save("recorded_obs_" + clip, obs=np.asarray(d.misc_data["obs"], dtype=np.float32),
                 shape_keys=np.array(list(shapes.keys())),
                 shape_vals=np.array([int(np.prod(shapes[k]["shape"] if isinstance(shapes[k], dict) else shapes[k])) for k in shapes]))
        out = ms_file.MSFileData(motion_data=d.motion_data, terrain_data=d.terrain_data, misc_data=misc)
        ms_file.save_ms_file(out, os.path.join(REPO, "data/motion_terrains", clip + ".pkl"))


# ----------------------------------------------------------------------------------
def rand_quat(g, n, small=False):
    q = torch.randn(n, 4, generator=g)
    if small:
        q[:, :3] *= 1e-4
        q[:, 3] = 1.0
    return tu.normalize(q)


def gen_quat_ops():
    g = torch.Generator().manual_seed(0)
    n = 256
    a = rand_quat(g, n)
    b = rand_quat(g, n)
    # edge rows: identity, w<0, tiny rotation, exactly opposite, identical
    a[0] = torch.tensor([0, 0, 0, 1.0]); b[0] = torch.tensor([0, 0, 0, 1.0])
    a[1] = torch.tensor([0, 0, 0, -1.0])
    a[2] = rand_quat(g, 1, small=True)[0]
    b[3] = -a[3]
    b[4] = a[4]
    b[5] = tu.normalize(a[5] + 1e-4 * torch.randn(4, generator=g))  # |sin| < 1e-3 slerp branch
    b[6] = tu.normalize(a[6] + 1e-6 * torch.randn(4, generator=g))
    a[7] = torch.tensor([1.0, 0, 0, 0]); a[8] = torch.tensor([0, 0, 1.0, 0])
    v = torch.randn(n, 3, generator=g)
    t = torch.rand(n, generator=g)
    t[9] = 0.0; t[10] = 1.0
    e = torch.randn(n, 3, generator=g)
    e[0] = 0.0
    e[1] = torch.tensor([1e-6, 0, 0])
    e[2] = torch.tensor([0, 0, 2e-5])
    e[3] = torch.tensor([3.5, 0, 0])  # > pi: normalize_angle wraps
    e[4] = torch.tensor([0, -6.0, 2.0])
    axis = tu.normalize(torch.randn(n, 3, generator=g))
    angle = (torch.rand(n, generator=g) - 0.5) * 8.0
    angle[0] = 0.0
    ax, ang = tu.quat_to_axis_angle(a)
    eax, eang = tu.exp_map_to_axis_angle(e)
    save("quat_ops", a=a, b=b, v=v, t=t, e=e, axis=axis, angle=angle,
         quat_mul=tu.quat_mul(a, b), quat_rotate=tu.quat_rotate(a, v),
         quat_conjugate=tu.quat_conjugate(a), quat_pos=tu.quat_pos(a),
         normalize=tu.normalize(v), q2aa_axis=ax, q2aa_angle=ang,
         aa2q=tu.axis_angle_to_quat(axis, angle), e2aa_axis=eax, e2aa_angle=eang,
         exp_map_to_quat=tu.exp_map_to_quat(e), quat_to_exp_map=tu.quat_to_exp_map(a),
         quat_diff=tu.quat_diff(a, b), quat_diff_angle=tu.quat_diff_angle(a, b),
         quat_normalize=tu.quat_normalize(a * 1.7), quat_to_tan_norm=tu.quat_to_tan_norm(a),
         slerp=tu.slerp(a, b, t), calc_heading=tu.calc_heading(a),
         calc_heading_quat_inv=tu.calc_heading_quat_inv(a),
         rotate_2d_vec=tu.rotate_2d_vec(v[:, :2], angle), normalize_angle=tu.normalize_angle(angle * 3.0),
         quat_multiply=tu.quat_multiply(a, b))


def load_char():
    m = kcm.KinCharModel(DEV)
    m.load_char_file(os.path.join(REF, "data/assets/humanoid.xml"))
    return m


def gen_char_model(m):
    jt = [int(j.joint_type.value) for j in m._joints]
    ax = np.zeros((15, 3), np.float32)
    for i, j in enumerate(m._joints):
        if j.axis is not None:
            ax[i] = npy(j.axis)
    save("char_model", parent=m._parent_indices, local_translation=m._local_translation,
         local_rotation=m._local_rotation, joint_type=np.array(jt), joint_axis=ax,
         dof_idx=np.array([j.dof_idx for j in m._joints]), dof_size=np.array(m._dof_size),
         lower=m._lower_dof_limits, upper=m._upper_dof_limits,
         body_names=np.array(m._body_names), contact_body_ids=m._contact_body_ids)


def gen_kin_ops(m):
    g = torch.Generator().manual_seed(1)
    n = 128
    dof = (torch.rand(n, 28, generator=g) - 0.5) * 3.0
    dof[0] = 0.0
    dof[1] = 1e-6
    dof[2, 0:3] = torch.tensor([0.0, 0.0, 5e-6])
    dof[3] *= 3.0  # > pi exp maps
    jr = m.dof_to_rot(dof)
    dof_back = m.rot_to_dof(jr)
    jr_rand = rand_quat(g, n * 14).reshape(n, 14, 4)
    jr_rand[0] = torch.tensor([0, 0, 0, 1.0])
    jr_rand[1] = rand_quat(g, 14, small=True)
    dof_rand = m.rot_to_dof(jr_rand)
    root_pos = torch.randn(n, 3, generator=g) * 3.0
    root_rot = rand_quat(g, n)
    bp, br = m.forward_kinematics(root_pos, root_rot, jr)
    bp2, br2 = m.forward_kinematics(root_pos, root_rot, jr_rand)
    jr1 = m.dof_to_rot(dof + 0.05 * torch.randn(n, 28, generator=g))
    dv = m.compute_dof_vel(jr, jr1, 1.0 / 30.0)
    save("kin_ops", dof=dof, joint_rot=jr, dof_back=dof_back, joint_rot_rand=jr_rand, dof_rand=dof_rand,
         root_pos=root_pos, root_rot=root_rot, body_pos=bp, body_rot=br, body_pos_rand=bp2, body_rot_rand=br2,
         joint_rot1=jr1, dof_vel=dv)


def make_mlib(m, clips, weights=None, tmp="/tmp/parc_golden"):
    os.makedirs(tmp, exist_ok=True)
    ypath = os.path.join(tmp, "motions.yaml")
    if weights is None:
        weights = [1.0 + 0.5 * i for i in range(len(clips))]
    with open(ypath, "w") as f:
        yaml.safe_dump({"motions": [{"file": os.path.join(REF, "data/motion_terrains", c + ".pkl"), "weight": w}
                                    for c, w in zip(clips, weights)]}, f)
    import parc.util.path_loader as pl
    pl.load_config = lambda p: yaml.safe_load(open(p).read())  # bypass $DATA_DIR assertion (SURVEY 8c iv)
    lib = ref_mlib.MotionLib(m, DEV, contact_info=True)
    lib._load_motion_file(ypath)
    return lib


def gen_motion_lib(m):
    lib = make_mlib(m, CLIPS)
    g = torch.Generator().manual_seed(2)
    n = 512
    ids = torch.randint(0, len(CLIPS), (n,), generator=g)
    lens = lib._motion_lengths[ids]
    t = (torch.rand(n, generator=g) * 1.3 - 0.1) * lens  # includes <0 and >len (CLAMP)
    t[0] = 0.0
    ids[1] = 0; t[1] = 0.10
    ids[2] = 0; t[2] = 0.21
    t[3] = lib._motion_lengths[ids[3]]
    i0, i1, bl = lib._calc_frame_blend(ids, t)
    rp, rr, rv, rav, jr, dv, ct = lib.calc_motion_frame(ids, t)
    bp, _ = m.forward_kinematics(rp, rr, jr)
    save("motion_lib", motion_weights=lib._motion_weights, motion_fps=lib._motion_fps, motion_dt=lib._motion_dt,
         motion_num_frames=lib._motion_num_frames, motion_lengths=lib._motion_lengths,
         motion_loop_modes=lib._motion_loop_modes, motion_start_idx=lib._motion_start_idx,
         motion_root_pos_delta=lib._motion_root_pos_delta,
         frame_root_pos=lib._frame_root_pos, frame_root_rot=lib._frame_root_rot,
         frame_root_vel=lib._frame_root_vel, frame_root_ang_vel=lib._frame_root_ang_vel,
         frame_joint_rot=lib._frame_joint_rot, frame_dof_vel=lib._frame_dof_vel, frame_contacts=lib._frame_contacts,
         q_ids=ids, q_times=t, idx0=i0, idx1=i1, blend=bl, root_pos=rp, root_rot=rr, root_vel=rv,
         root_ang_vel=rav, joint_rot=jr, dof_vel=dv, contacts=ct, body_pos=bp, dof_pos=m.rot_to_dof(jr))
    # WRAP variant of one clip (loop offset path, motion_lib.py:440)
    lib2 = make_mlib(m, ["civilization"])
    lib2._motion_loop_modes[:] = ref_mlib.LoopMode.WRAP.value
    ids2 = torch.zeros(64, dtype=torch.long)
    t2 = (torch.rand(64, generator=g) * 3.5 - 0.5) * lib2._motion_lengths[0]
    out = lib2.calc_motion_frame(ids2, t2)
    save("motion_lib_wrap", q_times=t2, root_pos=out[0], root_rot=out[1], joint_rot=out[4], contacts=out[6],
         root_vel=out[2], dof_vel=out[5])
    return lib


def gen_terrain(m):
    ray = geom_util.get_xy_points_cone(center=torch.zeros(2), dx=0.05, num_neg=2, num_pos=60,
                                       num_rays_neg=3, num_rays_pos=3, angle_between_rays=0.26179938779)
    d = ms_file.load_ms_file(os.path.join(REF, "data/motion_terrains/TEASER_TERRAIN.pkl"), load_misc=False)
    t = terrain_util.SubTerrain.from_ms_terrain_data(ref_file_io.MSTerrainData(**vars(d.terrain_data)), DEV)
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(4096, 2, generator=g) * 1.2 - 0.1) * 40.4 - 0.4
    # exact half-way cells exercise round-half-even
    k = torch.arange(0, 64, dtype=torch.float32)
    pts[:64, 0] = t.min_point[0] + (k + 0.5) * t.dxdy[0]
    pts[:64, 1] = t.min_point[1] + (k + 1.5) * t.dxdy[1]
    idx = t.get_grid_index(pts)
    vals = t.get_hf_val_from_points(pts)
    save("terrain_lookup", ray_points=ray, hf=t.hf, min_point=t.min_point, dxdy=t.dxdy, dims=t.dims,
         points=pts, grid_index=idx, hf_vals=vals)


def gen_terrain_slice(m):
    """terrain_util.slice_terrain_around_motion (terrain_util.py:1587-1660), as the recorder calls it
    (ig_parkour_env.py:715-720): padding = round(1.0 // dx) * dx, localize=True."""
    d = ms_file.load_ms_file(os.path.join(REF, "data/motion_terrains/TEASER_TERRAIN.pkl"), load_misc=False)
    t = terrain_util.SubTerrain.from_ms_terrain_data(ref_file_io.MSTerrainData(**vars(d.terrain_data)), DEV)
    g = torch.Generator().manual_seed(11)
    out = {}
    for i, (x0, y0, L) in enumerate([(3.0, 4.0, 40), (-1.0, 20.0, 90), (37.5, 37.9, 25)]):  # inside / over the low edge / over the high edge
        steps = 0.08 * torch.randn(L, 2, generator=g) + torch.tensor([0.05, 0.02])
        xy = torch.tensor([x0, y0]) + torch.cumsum(steps, 0)
        z = 0.8 + 0.05 * torch.randn(L, generator=g)
        frames = torch.cat([xy, z[:, None]], -1).to(torch.float32)
        padding = round(1.0 // t.dxdy[0].item()) * t.dxdy[0].item()
        st, loc = terrain_util.slice_terrain_around_motion(frames, t, padding=padding)
        out.update({f"frames{i}": frames, f"padding{i}": np.float64(padding), f"hf{i}": st.hf, f"hf_maxmin{i}": st.hf_maxmin,
                    f"min_point{i}": st.min_point, f"dims{i}": st.dims, f"local{i}": loc})
    save("terrain_slice", hf=t.hf, hf_maxmin=t.hf_maxmin, min_point=t.min_point, dxdy=t.dxdy, **out)


def gen_terrain_wide(m):
    """DeepMimicEnv.build_terrain_wide (dm_env.py:318-445): terrains stacked along x per motion and along y per copy."""
    cfg = env_config()
    dm = object.__new__(dm_env.DeepMimicEnv)
    dm._device = DEV
    dm._terrains_per_motion = 2
    dm._mlib = make_mlib(m, ["sfu", "civilization", "TEASER_TERRAIN", "dec2024_teaser_717_1_opt_dm"])
    _orig = terrain_util.SubTerrain.numpy_copy

    def _numpy_copy64(self):  # numpy>=2 drift, see build_harness
        t = _orig(self)
        t.min_point = t.min_point.astype(np.float64)
        return t

    terrain_util.SubTerrain.numpy_copy = _numpy_copy64
    try:
        dm.build_terrain_wide(cfg["env"], "/tmp/parc_golden/terrain_wide.pkl")
    finally:
        terrain_util.SubTerrain.numpy_copy = _orig
    save("terrain_wide", hf=dm._terrain.hf, min_point=dm._terrain.min_point, dxdy=dm._terrain.dxdy, dims=dm._terrain.dims,
         motion_offsets=dm._dm_motion_offsets, clips=np.array(["sfu", "civilization", "TEASER_TERRAIN", "dec2024_teaser_717_1_opt_dm"]))


# ----------------------------------------------------------------------------------
def env_config():
    cfg = yaml.safe_load(open(os.path.join(REF, "data/configs/tracker_config/dm_env_default.yaml")).read())
    return cfg


def build_harness(m, clips, num_envs, cfg):
    """Bare IGParkourEnv + DeepMimicEnv with the attributes their methods read."""
    env_config_ = cfg["env"]
    dm = object.__new__(dm_env.DeepMimicEnv)
    mdu.RefCharEnv.__init__(dm, cfg, num_envs, DEV, False, m)
    dm_cfg = env_config_["dm"]
    dm._random_reset_pos = False
    dm._min_motion_weight = dm_cfg.get("min_motion_weight", 0.01)
    dm._demo_mode = False
    dm._rand_reset = True
    dm._ignore_fail_rates = False
    dm._terrains_per_motion = 1
    dm._one_motion_mode = False
    dm._selected_motion_id = 0
    dm._terrain_build_mode = "square"
    dm._record_motion_frame_hist = False
    dm._mlib = make_mlib(m, clips)
    dm._motion_ids = torch.zeros(num_envs, dtype=torch.int64)
    dm._motion_terrain_ids = torch.zeros(num_envs, dtype=torch.int64)
    dm._motion_time_offsets = torch.zeros(num_envs, dtype=torch.float32)
    dm._motion_id_fail_rates = torch.ones(dm._mlib.num_motions(), dtype=torch.float32)
    dm._ema_weight = 0.01
    dm._motion_start_time_fraction = torch.zeros(num_envs)
    # numpy>=2 drift: `python_float - np.float32` now yields np.float32, which torch refuses to
    # assign (dm_env.py:222).  Under the reference's numpy 1.x it was a float64; restore that by
    # handing build_terrain_square float64 min_points (same values, double arithmetic, f32 store).
    _orig_numpy_copy = terrain_util.SubTerrain.numpy_copy

    def _numpy_copy64(self):
        t = _orig_numpy_copy(self)
        t.min_point = t.min_point.astype(np.float64)
        return t

    terrain_util.SubTerrain.numpy_copy = _numpy_copy64
    try:
        dm.build_terrain_square(env_config_, "/tmp/parc_golden/terrain.pkl")
    finally:
        terrain_util.SubTerrain.numpy_copy = _orig_numpy_copy

    env = object.__new__(ipe.IGParkourEnv)
    env._num_envs = num_envs
    env._device = DEV
    env._visualize = False
    env._num_dm_envs = num_envs
    env._dm_env = dm
    env._kin_char_model = m
    env._config = cfg
    env._timestep = 1.0 / env_config_["control_freq"]
    env._episode_length = env_config_["episode_length"]
    env._global_obs = env_config_["global_obs"]
    env._root_height_obs = env_config_.get("root_height_obs", True)
    env._enable_early_termination = env_config_["enable_early_termination"]
    env._termination_height = torch.tensor(env_config_["termination_height"], dtype=torch.float32)
    env._pose_termination = env_config_.get("pose_termination", False)
    env._pose_termination_dist = torch.tensor(env_config_["pose_termination_dist"], dtype=torch.float32)
    env._tar_obs_steps = torch.tensor(env_config_["tar_obs_steps"], dtype=torch.int)
    env._use_contact_info = bool(env_config_["use_contact_info"])
    env._contact_weights = torch.tensor(env_config_["contact_weights"], dtype=torch.float32)
    env._debug_visuals = False
    env._enable_tar_obs = bool(env_config_.get("enable_tar_obs", True))
    env._global_root_height_obs = env_config_["global_root_height_obs"]
    env._track_root = env_config_["track_root"]
    env._track_root_h = env_config_["track_root_h"]
    env._root_pos_termination_dist = env_config_["root_pos_termination_dist"]
    env._root_rot_termination_angle = env_config_["root_rot_termination_angle"]
    tw = sum(env_config_[k] for k in ["pose_w", "vel_w", "root_pos_w", "root_vel_w", "key_pos_w"])
    env._pose_w = env_config_["pose_w"] / tw
    env._vel_w = env_config_["vel_w"] / tw
    env._root_pos_w = env_config_["root_pos_w"] / tw
    env._root_vel_w = env_config_["root_vel_w"] / tw
    env._key_pos_w = env_config_["key_pos_w"] / tw
    env._report_tracking_error = True
    env._use_heightmap = True
    env._never_done = False
    env._start_compute_time = 0.0
    env._write_agent_states_flag = False
    env._ray_xy_points = geom_util.get_xy_points_cone(
        center=torch.zeros(2), dx=env_config_["ray_dx"], num_neg=env_config_["ray_points_behind"],
        num_pos=env_config_["ray_points_ahead"], num_rays_neg=env_config_["ray_num_left"],
        num_rays_pos=env_config_["ray_num_right"], angle_between_rays=env_config_["ray_angle"])
    env._ray_hfs = torch.zeros(num_envs, env._ray_xy_points.shape[0])
    # env offsets: ig_parkour_env.py:389-398
    spacing = env_config_["env_spacing"]
    per_row = int(np.sqrt(num_envs))
    env._env_offsets = torch.zeros(num_envs, 3)
    for i in range(num_envs):
        env._env_offsets[i, 0] = spacing * 2 * (i % per_row)
        env._env_offsets[i, 1] = spacing * 2 * (i // per_row)
    # sim tensors (what Isaac Gym would own)
    z = lambda *s: torch.zeros(*s, dtype=torch.float32)
    env._char_root_pos = z(num_envs, 3); env._char_root_rot = z(num_envs, 4)
    env._char_root_vel = z(num_envs, 3); env._char_root_ang_vel = z(num_envs, 3)
    env._char_dof_pos = z(num_envs, 28); env._char_dof_vel = z(num_envs, 28)
    env._char_rigid_body_pos = z(num_envs, 15, 3); env._char_rigid_body_rot = z(num_envs, 15, 4)
    env._char_rigid_body_vel = z(num_envs, 15, 3); env._char_rigid_body_ang_vel = z(num_envs, 15, 3)
    env._char_contact_forces = z(num_envs, 15, 3)
    env._ref_root_pos = z(num_envs, 3); env._ref_root_rot = z(num_envs, 4)
    env._ref_root_vel = z(num_envs, 3); env._ref_root_ang_vel = z(num_envs, 3)
    env._ref_body_pos = z(num_envs, 15, 3); env._ref_joint_rot = z(num_envs, 14, 4)
    env._ref_dof_pos = z(num_envs, 28); env._ref_dof_vel = z(num_envs, 28)
    env._ref_contacts = z(num_envs, 15)
    env._key_body_ids = torch.tensor([m.get_body_id(b) for b in env_config_["key_bodies"]], dtype=torch.long)
    env._contact_body_ids = torch.zeros(0, dtype=torch.long)
    env._parse_joint_err_weights(env_config_.get("joint_err_w", None))
    env._give_sim_tensor_views()
    env._reward_buf = z(num_envs)
    env._done_buf = torch.zeros(num_envs, dtype=torch.int)
    env._timestep_buf = torch.zeros(num_envs, dtype=torch.int)
    env._time_buf = z(num_envs)
    env._ep_num_buf = torch.zeros(num_envs, dtype=torch.int64)
    env._actors_need_reset = torch.zeros(num_envs, 1, dtype=torch.bool)
    env._info = dict()
    env._give_data_buffer_views()
    # row width (ig_parkour_env.py:911-958): char 136 (+1: root_height_obs leads the row, ig_char_env.py:620) | tar 6 x 105 | tar contacts 6 x 15 |
    # char contacts 15 | hf 441; `enable_tar_obs: false` drops the two target blocks, `use_contact_info: false` the two contact blocks
    env._obs_buf = z(num_envs, 136 + int(bool(env_config_["global_root_height_obs"])) + (630 if env._enable_tar_obs else 0)
                     + (90 if env._enable_tar_obs and env._use_contact_info else 0) + (15 if env._use_contact_info else 0) + 441)
    # Isaac Gym refresh calls are no-ops here: the caller injects the sim tensors
    env._refresh_sim_tensors = lambda: ipe.IGParkourEnv._refresh_obs_hfs(env)
    env.write_agent_states = lambda: None
    return env, dm


def inject_state(env, dm, m, g, noise=0.02, big_noise_rows=()):
    """char state = reference pose at the NEXT step's time + noise; rigid bodies = FK(state)."""
    n = env._num_envs
    t_next = (env._timestep_buf + 1).float() * env._timestep + dm._motion_time_offsets
    rp, rr, rv, rav, jr, dv, ct = dm._mlib.calc_motion_frame(dm._motion_ids, t_next)
    rp[..., 0:2] = dm._move_to_motion_terrain(rp[..., 0:2])
    dof = m.rot_to_dof(jr)
    sc = torch.full((n, 1), noise)
    for r in big_noise_rows:
        sc[r] = 0.6
    env._char_root_pos[:] = rp + sc * torch.randn(n, 3, generator=g)
    q = rr + sc * torch.randn(n, 4, generator=g)
    env._char_root_rot[:] = tu.normalize(q)
    env._char_root_vel[:] = rv + sc * torch.randn(n, 3, generator=g)
    env._char_root_ang_vel[:] = rav + sc * torch.randn(n, 3, generator=g)
    env._char_dof_pos[:] = dof + sc * torch.randn(n, 28, generator=g)
    env._char_dof_vel[:] = dv + 5 * sc * torch.randn(n, 28, generator=g)
    bp, br = m.forward_kinematics(env._char_root_pos, env._char_root_rot, m.dof_to_rot(env._char_dof_pos))
    env._char_rigid_body_pos[:] = bp
    env._char_rigid_body_rot[:] = br
    f = torch.randn(n, 15, 3, generator=g) * 0.8
    mask = torch.rand(n, 15, generator=g) < 0.5
    f[mask] = 0.0
    f[0, 0] = torch.tensor([3e-6, 0, 0]); f[0, 1] = torch.tensor([2e-5, 0, 0])
    env._char_contact_forces[:] = f


def state_dict(env, dm, prefix):
    d = {
        "char_root_pos": env._char_root_pos, "char_root_rot": env._char_root_rot,
        "char_root_vel": env._char_root_vel, "char_root_ang_vel": env._char_root_ang_vel,
        "char_dof_pos": env._char_dof_pos, "char_dof_vel": env._char_dof_vel,
        "char_body_pos": env._char_rigid_body_pos, "contact_forces": env._char_contact_forces,
        "motion_ids": dm._motion_ids, "terrain_ids": dm._motion_terrain_ids,
        "time_offsets": dm._motion_time_offsets, "timestep": env._timestep_buf, "time": env._time_buf,
        "fail_rates": dm._motion_id_fail_rates, "done": env._done_buf,
    }
    return {prefix + k: npy(v) for k, v in d.items()}


def out_dict(env, dm, prefix):
    d = {
        "ref_root_pos": env._ref_root_pos, "ref_root_rot": env._ref_root_rot, "ref_root_vel": env._ref_root_vel,
        "ref_root_ang_vel": env._ref_root_ang_vel, "ref_joint_rot": env._ref_joint_rot,
        "ref_dof_pos": env._ref_dof_pos, "ref_dof_vel": env._ref_dof_vel, "ref_body_pos": env._ref_body_pos,
        "ref_contacts": env._ref_contacts, "ray_hfs": env._ray_hfs, "obs": env._obs_buf,
        "reward": env._reward_buf, "done": env._done_buf, "timestep": env._timestep_buf, "time": env._time_buf,
        "fail_rates": dm._motion_id_fail_rates,
    }
    for k, v in env._info["rewards"].items():
        d["r_" + k] = v
    if "tracking_error" in env._info:
        d["tracking_error"] = env._info["tracking_error"]
    return {prefix + k: npy(v) for k, v in d.items()}


def gen_env_step(m):
    cfg = env_config()
    n = 64
    env, dm = build_harness(m, CLIPS, n, cfg)
    g = torch.Generator().manual_seed(4)
    terr = dm._terrain
    arrs = {"hf": terr.hf, "hf_min_point": terr.min_point, "hf_dxdy": terr.dxdy, "hf_dims": terr.dims,
            "motion_offsets": dm._dm_motion_offsets, "env_offsets": env._env_offsets,
            "ray_points": env._ray_xy_points, "joint_err_w": env._joint_err_w, "dof_err_w": env._dof_err_w,
            "clips": np.array(CLIPS), "weights": dm._mlib._motion_weights}

    # ---- reset fixture (DeepMimicEnv.reset with torch RNG; samples recorded for injection)
    torch.manual_seed(123)
    ids = torch.arange(n)
    dm.reset(ids)
    arrs.update({k: npy(v) for k, v in {"reset_motion_ids": dm._motion_ids, "reset_terrain_ids": dm._motion_terrain_ids,
                 "reset_time_offsets": dm._motion_time_offsets,
                 "reset_xy_noise": env._char_root_pos[:, 0:2] - env._ref_root_pos[:, 0:2],
                 "reset_char_root_pos": env._char_root_pos, "reset_char_root_rot": env._char_root_rot,
                 "reset_char_root_vel": env._char_root_vel, "reset_char_root_ang_vel": env._char_root_ang_vel,
                 "reset_char_dof_pos": env._char_dof_pos, "reset_char_dof_vel": env._char_dof_vel,
                 "reset_ref_root_pos": env._ref_root_pos, "reset_ref_dof_pos": env._ref_dof_pos,
                 "reset_ref_contacts": env._ref_contacts}.items()})
    env._refresh_sim_tensors()
    env._update_observations(ids)
    arrs["reset_obs"] = npy(env._obs_buf)
    arrs["reset_ray_hfs"] = npy(env._ray_hfs)

    # ---- step fixtures: 3 consecutive control steps on injected state
    env._timestep_buf[:] = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32)
    env._timestep_buf[0:4] = 0  # first step after reset (time > 1e-5 holds at ts=1)
    dm._motion_time_offsets[4] = dm._mlib._motion_lengths[dm._motion_ids[4]] - 0.01  # motion end -> FAIL
    env._timestep_buf[5] = 299  # time >= episode_length -> TIME
    for s in range(3):
        inject_state(env, dm, m, g, noise=0.02, big_noise_rows=(6, 7, 8, 9))
        if s == 0:
            env._char_rigid_body_pos[10, 5] += torch.tensor([0.0, 0.0, 0.8])  # single-body pose fail
        arrs.update(state_dict(env, dm, f"s{s}_in_"))
        ig_env.IGEnv._post_physics_step(env)
        arrs.update(out_dict(env, dm, f"s{s}_out_"))
    save("env_step", **arrs)


def gen_env_step_fall(m):
    """`contact_bodies: [right_foot, left_foot]` (mgdm_dm_util.py:349-360): a fall = a contact force above 0.1 on some body other
    than those AND some such body lower than `termination_height` above the terrain under it (dm_env.py:628-635 looks the terrain
    up per body).  Same scene and reset as env_step.npz; one step on injected state with the four combinations spread over the rows."""
    cfg = env_config()
    cfg["env"]["pose_termination"] = False   # isolate the fall rule: moving a body to a chosen height would trip the pose rule as well
    n = 64
    env, dm = build_harness(m, CLIPS, n, cfg)
    feet = [m.get_body_id("right_foot"), m.get_body_id("left_foot")]
    env._contact_body_ids = torch.tensor(feet, dtype=torch.long)
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(321)
    dm.reset(torch.arange(n))
    env._refresh_sim_tensors()
    env._update_observations(torch.arange(n))
    env._timestep_buf[:] = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
    inject_state(env, dm, m, g, noise=0.02)
    # ground height under every body (the same lookup update_done does) to place bodies relative to it
    gpos = env._char_rigid_body_pos[..., 0:2] + env._env_offsets[:, 0:2].unsqueeze(1)
    inds = dm._terrain.get_grid_index(gpos)
    ground = dm._terrain.hf[inds[..., 0], inds[..., 1]]
    f = torch.zeros(n, 15, 3)
    for e in range(n):
        kind = e % 4
        # feet may always push (they are contact bodies): must never count
        f[e, feet[0]] = torch.tensor([0.0, 0.0, 300.0]); f[e, feet[1]] = torch.tensor([5.0, -3.0, 200.0])
        b_f, b_h = 1 + (e % 8), 3 + (e % 6)          # bodies 1..8 / 3..8: never a foot
        if kind in (1, 3):                            # contact on a non-contact body (one component just above / below 0.1)
            f[e, b_f] = torch.tensor([0.0, 0.1001 if e % 8 < 4 else -0.35, 0.0])
        if kind == 0:
            f[e, b_f] = torch.tensor([0.0999, -0.0999, 0.05])   # below the threshold on every component: no contact
        if kind in (2, 3):                            # some non-contact body lower than termination_height above its ground
            env._char_rigid_body_pos[e, b_h, 2] = ground[e, b_h] - env._env_offsets[e, 2] * 0 + 0.149
        else:
            env._char_rigid_body_pos[e, 1:, 2] = torch.maximum(env._char_rigid_body_pos[e, 1:, 2], ground[e, 1:] + 0.151)
            env._char_rigid_body_pos[e, feet, 2] = ground[e, feet] + 0.02   # low feet never count
    env._char_contact_forces[:] = f
    arrs = {"contact_body_ids": np.array(feet, np.int64), "termination_height": np.float32(cfg["env"]["termination_height"]),
            "pose_termination": np.int32(0)}
    arrs.update(state_dict(env, dm, "in_"))
    ig_env.IGEnv._post_physics_step(env)
    arrs.update(out_dict(env, dm, "out_"))
    save("env_step_fall", **arrs)
    d = npy(env._done_buf)
    print("fall fixture: done histogram", np.bincount(d, minlength=4), "rows kind 3 failed:", d[3::4])


def gen_env_step_local_root(m):
    """`track_root: False` (mgdm_dm_util.py:294-310, 386): the reward drops the horizontal root position error and compares root rotation,
    root velocities and key positions in each character's own heading frame (convert_to_local :247-267); compute_done skips the root
    position / rotation termination.  Same scene and reset as env_step.npz; one step on injected state, a few rows turned far off the
    reference heading so that the two frames differ."""
    cfg = env_config()
    cfg["env"]["track_root"] = False
    n = 64
    env, dm = build_harness(m, CLIPS, n, cfg)
    assert env._track_root is False
    g = torch.Generator().manual_seed(23)
    torch.manual_seed(77)
    dm.reset(torch.arange(n))
    env._refresh_sim_tensors()
    env._update_observations(torch.arange(n))
    env._timestep_buf[:] = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
    inject_state(env, dm, m, g, noise=0.02, big_noise_rows=(6, 7, 8, 9))
    # rows 16..31: the character turned about z by up to +-2.5 rad and displaced by metres: with track_root the episode would end
    ang = (torch.rand(16, generator=g) * 5.0 - 2.5)
    dq = torch.stack([torch.zeros(16), torch.zeros(16), torch.sin(0.5 * ang), torch.cos(0.5 * ang)], dim=-1)
    env._char_root_rot[16:32] = tu.quat_mul(dq, env._char_root_rot[16:32])
    shift = torch.zeros(n, 3); shift[24:40, 0:2] = torch.rand(16, 2, generator=g) * 4.0 - 2.0
    env._char_root_pos[:] = env._char_root_pos + shift
    bp, br = m.forward_kinematics(env._char_root_pos, env._char_root_rot, m.dof_to_rot(env._char_dof_pos))
    env._char_rigid_body_pos[:] = bp
    env._char_rigid_body_rot[:] = br
    arrs = {"track_root": np.int32(0)}
    arrs.update(state_dict(env, dm, "in_"))
    ig_env.IGEnv._post_physics_step(env)
    arrs.update(out_dict(env, dm, "out_"))
    save("env_step_local_root", **arrs)
    print("local-root fixture: done histogram", np.bincount(npy(env._done_buf), minlength=4), "reward range", float(env._reward_buf.min()), float(env._reward_buf.max()))


def gen_env_step_reward_done_switches(m):
    """`track_root_h: False` (compute_deepmimic_reward: the vertical root error leaves the root-position term, mgdm_dm_util.py) and
    `enable_early_termination: False` (compute_done: only the time limit and the motion end finish an episode).  Two sub-fixtures on the state of
    env_step_local_root (sixteen characters turned, sixteen displaced) with sixteen more rows lifted / lowered by up to 0.5 m, so that both switches
    change what the default config would give: keys `h0_*` (track_root_h off) and `et0_*` (early termination off)."""
    arrs = {}
    for tag, key, val in (("h0_", "track_root_h", False), ("et0_", "enable_early_termination", False)):
        cfg = env_config()
        cfg["env"][key] = val
        n = 64
        env, dm = build_harness(m, CLIPS, n, cfg)
        assert getattr(env, "_" + key) is val
        g = torch.Generator().manual_seed(29)
        torch.manual_seed(78)
        dm.reset(torch.arange(n))
        env._refresh_sim_tensors()
        env._update_observations(torch.arange(n))
        env._timestep_buf[:] = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
        inject_state(env, dm, m, g, noise=0.02, big_noise_rows=(6, 7, 8, 9))
        ang = (torch.rand(16, generator=g) * 5.0 - 2.5)
        dq = torch.stack([torch.zeros(16), torch.zeros(16), torch.sin(0.5 * ang), torch.cos(0.5 * ang)], dim=-1)
        env._char_root_rot[16:32] = tu.quat_mul(dq, env._char_root_rot[16:32])
        shift = torch.zeros(n, 3); shift[24:40, 0:2] = torch.rand(16, 2, generator=g) * 4.0 - 2.0
        shift[40:56, 2] = torch.rand(16, generator=g) * 1.0 - 0.5          # lifted / lowered: what track_root_h decides about
        env._char_root_pos[:] = env._char_root_pos + shift
        bp, br = m.forward_kinematics(env._char_root_pos, env._char_root_rot, m.dof_to_rot(env._char_dof_pos))
        env._char_rigid_body_pos[:] = bp
        env._char_rigid_body_rot[:] = br
        arrs.update(state_dict(env, dm, tag + "in_"))
        ig_env.IGEnv._post_physics_step(env)
        arrs.update(out_dict(env, dm, tag + "out_"))
        print(tag, "done histogram", np.bincount(npy(env._done_buf), minlength=4), "reward range", float(env._reward_buf.min()), float(env._reward_buf.max()))
    save("env_step_reward_done_switches", **arrs)


def gen_env_step_root_height_obs(m):
    """`global_root_height_obs: True` (ig_parkour_env.py:84, passed to compute_char_obs as root_height_obs, :904): the root height is one
    more observation in front of the character block (ig_char_env.py:620-622) -- 1 313 columns.  With and without `global_obs`."""
    for gl in (False, True):
        cfg = env_config()
        cfg["env"]["global_root_height_obs"] = True
        cfg["env"]["global_obs"] = gl
        n = 64
        env, dm = build_harness(m, CLIPS, n, cfg)
        g = torch.Generator().manual_seed(41)
        torch.manual_seed(55)
        dm.reset(torch.arange(n))
        env._refresh_sim_tensors()
        env._update_observations(torch.arange(n))
        assert env._obs_buf.shape[1] == 1313
        arrs = {"global_obs": np.int32(gl), "global_root_height_obs": np.int32(1), "reset_obs": npy(env._obs_buf)}
        arrs.update(state_dict(env, dm, "reset_"))
        env._timestep_buf[:] = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
        inject_state(env, dm, m, g, noise=0.02, big_noise_rows=(6, 7, 8, 9))
        arrs.update(state_dict(env, dm, "in_"))
        ig_env.IGEnv._post_physics_step(env)
        arrs.update(out_dict(env, dm, "out_"))
        save("env_step_root_height_obs" + ("_global" if gl else ""), **arrs)


def gen_env_step_global_obs(m):
    """`global_obs: True` (ig_parkour_env.py:83): compute_char_obs (ig_char_env.py:586-589, :603) keeps the root rotation, the root
    velocities and the key-body offsets in the global frame; compute_tar_obs (mgdm_dm_util.py:417) keeps the targets' root offset, root
    rotation and key offsets global and does NOT add the root offset to the key offsets.  Same scene and reset as env_step.npz; the reset
    observation and one step on injected state."""
    cfg = env_config()
    cfg["env"]["global_obs"] = True
    n = 64
    env, dm = build_harness(m, CLIPS, n, cfg)
    assert env._global_obs is True
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(99)
    dm.reset(torch.arange(n))
    env._refresh_sim_tensors()
    env._update_observations(torch.arange(n))
    arrs = {"global_obs": np.int32(1), "reset_obs": npy(env._obs_buf)}
    arrs.update(state_dict(env, dm, "reset_"))
    env._timestep_buf[:] = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
    inject_state(env, dm, m, g, noise=0.02, big_noise_rows=(6, 7, 8, 9))
    arrs.update(state_dict(env, dm, "in_"))
    ig_env.IGEnv._post_physics_step(env)
    arrs.update(out_dict(env, dm, "out_"))
    save("env_step_global_obs", **arrs)


def gen_env_step_obs_blocks(m):
    """`use_contact_info: false` (ig_parkour_env.py:72-73: no contact blocks in the observation :927-946, no contact term in the reward
    :1032-1040) and `enable_tar_obs: false` (:83: no target block, mgdm_dm_util.py:482-493, and no target-contact block :928-930), each alone
    and both together.  Same scene and reset as env_step.npz; the reset observation and one step on injected state."""
    for uci, eto in ((False, True), (True, False), (False, False)):
        cfg = env_config()
        cfg["env"]["use_contact_info"] = uci
        cfg["env"]["enable_tar_obs"] = eto
        n = 64
        env, dm = build_harness(m, CLIPS, n, cfg)
        assert env._use_contact_info is uci and env._enable_tar_obs is eto
        g = torch.Generator().manual_seed(61)
        torch.manual_seed(67)
        dm.reset(torch.arange(n))
        env._refresh_sim_tensors()
        env._update_observations(torch.arange(n))
        arrs = {"use_contact_info": np.int32(uci), "enable_tar_obs": np.int32(eto), "reset_obs": npy(env._obs_buf)}
        arrs.update(state_dict(env, dm, "reset_"))
        env._timestep_buf[:] = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
        inject_state(env, dm, m, g, noise=0.02, big_noise_rows=(6, 7, 8, 9))
        arrs.update(state_dict(env, dm, "in_"))
        ig_env.IGEnv._post_physics_step(env)
        arrs.update(out_dict(env, dm, "out_"))
        save("env_step_obs_blocks_c%d_t%d" % (int(uci), int(eto)), **arrs)
        print("obs blocks fixture", uci, eto, "obs width", env._obs_buf.shape[1], "reward terms", sorted(env._info["rewards"].keys()))


def gen_env_step_far(m):
    """The 64-env scene of env_step.npz with the env origins moved out to where the envs of a 65 536-env run sit (ig_parkour_env.py:389-398:
    env_spacing * 2 * column, up to ~1 km): rows 0..31 by (+300 m, +300 m), rows 32..63 by (+1000 m, +700 m).  The env-local root
    positions (dm_env.py:554-558: motion position + terrain offset - env offset) are then hundreds of metres, where one fp32 ulp is
    3e-5 .. 6e-5 m: this fixture is what the REFERENCE's own arithmetic produces there -- the large-N allowance of the GPU parity tests
    (1e-5 + 2 ulp of the coordinates) is checked against it instead of argued."""
    cfg = env_config()
    n = 64
    env, dm = build_harness(m, CLIPS, n, cfg)
    env._env_offsets[:32, 0] += 300.0; env._env_offsets[:32, 1] += 300.0       # in place: the DeepMimic env holds a view of the same tensor
    env._env_offsets[32:, 0] += 1000.0; env._env_offsets[32:, 1] += 700.0
    assert dm._env_offsets.data_ptr() == env._env_offsets.data_ptr() and torch.equal(dm._env_offsets, env._env_offsets)
    g = torch.Generator().manual_seed(71)
    torch.manual_seed(73)
    dm.reset(torch.arange(n))
    env._refresh_sim_tensors()
    env._update_observations(torch.arange(n))
    arrs = {"env_offsets": npy(env._env_offsets), "reset_obs": npy(env._obs_buf)}
    arrs.update(state_dict(env, dm, "reset_"))
    env._timestep_buf[:] = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
    inject_state(env, dm, m, g, noise=0.02, big_noise_rows=(6, 7, 8, 9))
    arrs.update(state_dict(env, dm, "in_"))
    ig_env.IGEnv._post_physics_step(env)
    arrs.update(out_dict(env, dm, "out_"))
    save("env_step_far", **arrs)
    print("far fixture: |root| max", float(env._char_root_pos.abs().max()), "done histogram", np.bincount(npy(env._done_buf), minlength=4))


def gen_action_bounds(m):
    """IGCharEnv._build_action_bounds_pd (ig_char_env.py:307-347) on a bare instance: Isaac Gym's get_actor_dof_properties is replaced by the
    joint limits the reference's own KinCharModel parsed from the MJCF (radians; what Isaac Gym reports for the same asset)."""
    obj = object.__new__(ig_char_env.IGCharEnv)
    obj._envs = [None]
    obj._kin_char_model = m
    obj._get_char_actor_handle = lambda: 0
    lo, hi = npy(m._lower_dof_limits).astype(np.float32), npy(m._upper_dof_limits).astype(np.float32)
    obj._gym = types.SimpleNamespace(get_actor_dof_properties=lambda env_handle, char_handle: {"lower": lo, "upper": hi})
    low, high = ig_char_env.IGCharEnv._build_action_bounds_pd(obj)
    save("action_bounds", dof_lower=lo, dof_upper=hi, action_low=np.asarray(low), action_high=np.asarray(high))


def gen_done_table():
    g = torch.Generator().manual_seed(5)
    n = 128
    done_buf = torch.zeros(n, dtype=torch.int)
    time = torch.rand(n, generator=g) * 12.0
    time[0] = 0.0; time[1] = 1e-6; time[2] = 10.0
    rr = rand_quat(g, n); trr = tu.normalize(rr + 0.3 * torch.randn(n, 4, generator=g))
    bp = torch.randn(n, 15, 3, generator=g)
    tbp = bp + 0.35 * torch.randn(n, 15, 3, generator=g)
    ptd = torch.tensor(env_config()["env"]["pose_termination_dist"], dtype=torch.float32)
    done = mdu.compute_done(done_buf=done_buf, time=time, ep_len=10.0, root_rot=rr, body_pos=bp,
                            char_root_pos=bp[:, 0], tar_root_rot=trr, tar_body_pos=tbp,
                            contact_force=torch.zeros(n, 15, 3), contact_body_ids=torch.zeros(0, dtype=torch.long),
                            termination_heights=torch.zeros(n, 15), pose_termination=True,
                            pose_termination_dist=ptd, global_obs=False, enable_early_termination=True,
                            track_root=True, root_pos_termination_dist=0.6, root_rot_termination_angle=1.309)
    tc = torch.rand(n, 15, generator=g)
    cf = torch.randn(n, 15, 3, generator=g)
    cw = torch.full((15,), 5.0)
    cr = mdu.compute_contact_reward(tc, cf, cw)
    save("done_table", time=time, root_rot=rr, tar_root_rot=trr, body_pos=bp, tar_body_pos=tbp,
         pose_termination_dist=ptd, done=done, tar_contacts=tc, contact_forces=cf, contact_r=cr)


if __name__ == "__main__":
    if len(sys.argv) > 1:   # only the named fixtures, e.g. `make_golden.py env_step_far action_bounds` (the others stay as committed)
        model = load_char()
        for name in sys.argv[1:]:
            globals()["gen_" + name](model)
        sys.exit(0)
    export_data_files()
    gen_quat_ops()
    model = load_char()
    gen_char_model(model)
    gen_kin_ops(model)
    gen_motion_lib(model)
    gen_terrain(model)
    gen_terrain_slice(model)
    gen_terrain_wide(model)
    gen_done_table()
    gen_env_step(model)
    gen_env_step_fall(model)
    gen_env_step_local_root(model)
    gen_env_step_reward_done_switches(model)
    gen_env_step_global_obs(model)
    gen_env_step_root_height_obs(model)
    gen_env_step_obs_blocks(model)
    gen_env_step_far(model)
    gen_action_bounds(model)
