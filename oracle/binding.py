"""ctypes binding of the CPU oracle (``oracle/parc_oracle.c``).

TEST INFRASTRUCTURE ONLY.  May be imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — never by anything under ``parc_amd/``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PARC_ORACLE_LIB selects another build of the same sources (e.g. libparc_oracle_asan.so from `make asan`, run with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so)): sanitizer runs are CPU-only
_LIB_PATH = os.environ.get("PARC_ORACLE_LIB", os.path.join(_HERE, "libparc_oracle.so"))

MAXB, MAXD, MAXS, MAXK = 16, 48, 8, 8
f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("parc_oracle.c", "parc_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OrcChar(C.Structure):
    _fields_ = [("num_bodies", C.c_int), ("dof_size", C.c_int), ("parent", C.c_int * MAXB),
                ("local_translation", (C.c_float * 3) * MAXB), ("local_rotation", (C.c_float * 4) * MAXB),
                ("joint_type", C.c_int * MAXB), ("joint_axis", (C.c_float * 3) * MAXB), ("dof_idx", C.c_int * MAXB)]


class OrcMotionLib(C.Structure):
    _fields_ = [("num_motions", C.c_int), ("num_frames_total", C.c_int64), ("num_joints", C.c_int),
                ("num_bodies", C.c_int),
                ("motion_weights", f32p), ("motion_fps", f32p), ("motion_dt", f32p), ("motion_lengths", f32p),
                ("motion_root_pos_delta", f32p), ("motion_num_frames", i64p), ("motion_start_idx", i64p),
                ("motion_loop_modes", i32p),
                ("frame_root_pos", f32p), ("frame_root_rot", f32p), ("frame_root_vel", f32p),
                ("frame_root_ang_vel", f32p), ("frame_joint_rot", f32p), ("frame_dof_vel", f32p),
                ("frame_contacts", f32p), ("dof_size", C.c_int)]


class OrcTerrain(C.Structure):
    _fields_ = [("hf", f32p), ("dims", C.c_int64 * 2), ("min_point", C.c_float * 2), ("dxdy", C.c_float * 2)]


class OrcEnvCfg(C.Structure):
    _fields_ = [("num_envs", C.c_int), ("num_key", C.c_int), ("key_body_ids", C.c_int * MAXK),
                ("num_tar_steps", C.c_int), ("tar_obs_steps", C.c_int * MAXS),
                ("num_rays", C.c_int), ("ray_points", f32p), ("timestep", C.c_float), ("timestep_d", C.c_double),
                ("episode_length", C.c_float), ("min_obs_h", C.c_float), ("max_obs_h", C.c_float),
                ("pose_w", C.c_float), ("vel_w", C.c_float), ("root_pos_w", C.c_float), ("root_vel_w", C.c_float),
                ("key_pos_w", C.c_float), ("joint_err_w", f32p), ("dof_err_w", f32p), ("contact_weights", f32p),
                ("pose_termination_dist", f32p), ("root_pos_termination_dist", C.c_float),
                ("root_rot_termination_angle", C.c_float), ("enable_early_termination", C.c_int),
                ("pose_termination", C.c_int), ("track_root", C.c_int), ("track_root_h", C.c_int),
                ("ema_weight", C.c_float), ("env_offsets", f32p), ("motion_offsets", f32p),
                ("terrains_per_motion", C.c_int), ("num_contact_bodies", C.c_int), ("contact_body_ids", C.c_int * 16),
                ("termination_height", C.c_float), ("global_obs", C.c_int), ("no_contact_info", C.c_int), ("no_tar_obs", C.c_int),
                ("root_height_obs", C.c_int)]


_STATE_F32 = ["char_root_pos", "char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos",
              "char_dof_vel", "char_body_pos", "contact_forces"]
_STATE_REF = ["ref_root_pos", "ref_root_rot", "ref_root_vel", "ref_root_ang_vel", "ref_joint_rot", "ref_dof_pos",
              "ref_dof_vel", "ref_body_pos", "ref_contacts"]
_STATE_OUT = ["ray_hfs", "obs", "reward", "reward_terms", "tracking_error"]


class OrcEnvState(C.Structure):
    _fields_ = ([(n, f32p) for n in _STATE_F32] + [("motion_ids", i64p), ("terrain_ids", i64p), ("time_offsets", f32p),
                ("time_buf", f32p), ("timestep_buf", i32p)] + [(n, f32p) for n in _STATE_REF] +
                [(n, f32p) for n in _STATE_OUT] + [("done", i32p), ("fail_rates", f32p)])


def _fp(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(f32p)


def _ip64(a):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(i64p)


def _ip32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(i32p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """Thin numpy front-end over the C oracle."""

    def __init__(self):
        self.lib = C.CDLL(build())
        self.lib.orc_mlib_create.restype = C.POINTER(OrcMotionLib)

    # ---- a1 -----------------------------------------------------------------------------
    def _unary(self, fn, a, out_dim):
        a = f32(a); n = a.shape[0]
        out = np.empty((n, out_dim) if out_dim > 1 else (n,), np.float32)
        getattr(self.lib, fn)(_fp(a), _fp(out), C.c_int(n))
        return out

    def _binary(self, fn, a, b, out_dim):
        a = f32(a); b = f32(b); n = a.shape[0]
        out = np.empty((n, out_dim) if out_dim > 1 else (n,), np.float32)
        getattr(self.lib, fn)(_fp(a), _fp(b), _fp(out), C.c_int(n))
        return out

    def quat_mul(self, a, b): return self._binary("orc_quat_mul", a, b, 4)
    def quat_multiply(self, a, b): return self._binary("orc_quat_multiply", a, b, 4)
    def quat_rotate(self, q, v): return self._binary("orc_quat_rotate", q, v, 3)
    def quat_conjugate(self, q): return self._unary("orc_quat_conjugate", q, 4)
    def quat_pos(self, q): return self._unary("orc_quat_pos", q, 4)
    def exp_map_to_quat(self, e): return self._unary("orc_exp_map_to_quat", e, 4)
    def quat_to_exp_map(self, q): return self._unary("orc_quat_to_exp_map", q, 3)
    def quat_diff(self, a, b): return self._binary("orc_quat_diff", a, b, 4)
    def quat_diff_angle(self, a, b): return self._binary("orc_quat_diff_angle", a, b, 1)
    def quat_normalize(self, q): return self._unary("orc_quat_normalize", q, 4)
    def quat_to_tan_norm(self, q): return self._unary("orc_quat_to_tan_norm", q, 6)
    def calc_heading(self, q): return self._unary("orc_calc_heading", q, 1)
    def calc_heading_quat_inv(self, q): return self._unary("orc_calc_heading_quat_inv", q, 4)
    def normalize_angle(self, x): return self._unary("orc_normalize_angle", x, 1)
    def rotate_2d_vec(self, v, ang): return self._binary("orc_rotate_2d_vec", v, ang, 2)
    def axis_angle_to_quat(self, axis, ang): return self._binary("orc_axis_angle_to_quat", axis, ang, 4)

    def normalize(self, x):
        x = f32(x); out = np.empty_like(x)
        self.lib.orc_normalize(_fp(x), _fp(out), C.c_int(x.shape[0]), C.c_int(x.shape[1]))
        return out

    def quat_to_axis_angle(self, q):
        q = f32(q); n = q.shape[0]
        ax = np.empty((n, 3), np.float32); an = np.empty(n, np.float32)
        self.lib.orc_quat_to_axis_angle(_fp(q), _fp(ax), _fp(an), C.c_int(n))
        return ax, an

    def exp_map_to_axis_angle(self, e):
        e = f32(e); n = e.shape[0]
        ax = np.empty((n, 3), np.float32); an = np.empty(n, np.float32)
        self.lib.orc_exp_map_to_axis_angle(_fp(e), _fp(ax), _fp(an), C.c_int(n))
        return ax, an

    def slerp(self, q0, q1, t):
        q0 = f32(q0); q1 = f32(q1); t = f32(t); out = np.empty_like(q0)
        self.lib.orc_slerp(_fp(q0), _fp(q1), _fp(t), _fp(out), C.c_int(q0.shape[0]))
        return out

    # ---- character ----------------------------------------------------------------------
    @staticmethod
    def make_char(parent, local_translation, local_rotation, joint_type, joint_axis, dof_idx, dof_size):
        c = OrcChar()
        nb = len(parent)
        c.num_bodies = nb; c.dof_size = int(dof_size)
        for b in range(nb):
            c.parent[b] = int(parent[b]); c.joint_type[b] = int(joint_type[b]); c.dof_idx[b] = int(dof_idx[b])
            for k in range(3):
                c.local_translation[b][k] = float(local_translation[b][k]); c.joint_axis[b][k] = float(joint_axis[b][k])
            for k in range(4):
                c.local_rotation[b][k] = float(local_rotation[b][k])
        return c

    def dof_to_rot(self, c, dof):
        dof = f32(dof); n = dof.shape[0]
        out = np.empty((n, c.num_bodies - 1, 4), np.float32)
        self.lib.orc_dof_to_rot(C.byref(c), _fp(dof), _fp(out), C.c_int(n))
        return out

    def rot_to_dof(self, c, jr):
        jr = f32(jr); n = jr.shape[0]
        out = np.empty((n, c.dof_size), np.float32)
        self.lib.orc_rot_to_dof(C.byref(c), _fp(jr), _fp(out), C.c_int(n))
        return out

    def forward_kinematics(self, c, root_pos, root_rot, jr):
        root_pos = f32(root_pos); root_rot = f32(root_rot); jr = f32(jr); n = jr.shape[0]
        bp = np.empty((n, c.num_bodies, 3), np.float32); br = np.empty((n, c.num_bodies, 4), np.float32)
        self.lib.orc_forward_kinematics(C.byref(c), _fp(root_pos), _fp(root_rot), _fp(jr), _fp(bp), _fp(br), C.c_int(n))
        return bp, br

    def compute_dof_vel(self, c, jr0, jr1, dt):
        jr0 = f32(jr0); jr1 = f32(jr1); n = jr0.shape[0]
        out = np.empty((n, c.dof_size), np.float32)
        self.lib.orc_compute_dof_vel(C.byref(c), _fp(jr0), _fp(jr1), C.c_float(dt), _fp(out), C.c_int(n))
        return out

    # ---- motion lib ---------------------------------------------------------------------
    def mlib_create(self, c, clips, weights):
        """clips: list of dict(root_pos, root_rot, joint_rot, contacts|None, fps, loop_mode int)."""
        nf = np.array([cl["root_pos"].shape[0] for cl in clips], np.int64)
        fps = np.array([cl["fps"] for cl in clips], np.int32)
        lm = np.array([cl["loop_mode"] for cl in clips], np.int32)
        w = np.array(weights, np.float64)
        rp = f32(np.concatenate([cl["root_pos"] for cl in clips]))
        rr = f32(np.concatenate([cl["root_rot"] for cl in clips]))
        jr = f32(np.concatenate([cl["joint_rot"] for cl in clips]))
        nb = c.num_bodies
        ct = f32(np.concatenate([cl["contacts"] if cl.get("contacts") is not None else
                                 np.zeros((cl["root_pos"].shape[0], nb), np.float32) for cl in clips]))
        ptr = self.lib.orc_mlib_create(C.byref(c), C.c_int(len(clips)), _ip64(nf), _ip32(fps), _ip32(lm),
                                       w.ctypes.data_as(C.POINTER(C.c_double)), _fp(rp), _fp(rr), _fp(jr), _fp(ct))
        return ptr

    @staticmethod
    def mlib_array(lib_ptr, name, shape, dtype=np.float32):
        p = getattr(lib_ptr.contents, name)
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(shape).astype(dtype).copy()

    def mlib_calc_frame_blend(self, lib, ids, times):
        ids = np.ascontiguousarray(ids, np.int64); times = f32(times); n = len(ids)
        i0 = np.empty(n, np.int64); i1 = np.empty(n, np.int64); bl = np.empty(n, np.float32)
        self.lib.orc_mlib_calc_frame_blend(lib, _ip64(ids), _fp(times), C.c_int(n), _ip64(i0), _ip64(i1), _fp(bl))
        return i0, i1, bl

    def mlib_calc_motion_frame(self, lib, ids, times):
        ids = np.ascontiguousarray(ids, np.int64); times = f32(times); n = len(ids)
        L = lib.contents
        J, B, D = L.num_joints, L.num_bodies, L.dof_size
        o = dict(root_pos=np.empty((n, 3), np.float32), root_rot=np.empty((n, 4), np.float32),
                 root_vel=np.empty((n, 3), np.float32), root_ang_vel=np.empty((n, 3), np.float32),
                 joint_rot=np.empty((n, J, 4), np.float32), dof_vel=np.empty((n, D), np.float32),
                 contacts=np.empty((n, B), np.float32))
        self.lib.orc_mlib_calc_motion_frame(lib, _ip64(ids), _fp(times), C.c_int(n), _fp(o["root_pos"]),
                                            _fp(o["root_rot"]), _fp(o["root_vel"]), _fp(o["root_ang_vel"]),
                                            _fp(o["joint_rot"]), _fp(o["dof_vel"]), _fp(o["contacts"]))
        return o

    # ---- terrain ------------------------------------------------------------------------
    @staticmethod
    def make_terrain(hf, min_point, dxdy):
        hf = f32(hf)
        t = OrcTerrain()
        t._keep = hf
        t.hf = _fp(hf)
        t.dims[0], t.dims[1] = hf.shape
        t.min_point[0], t.min_point[1] = float(np.float32(min_point[0])), float(np.float32(min_point[1]))
        t.dxdy[0], t.dxdy[1] = float(np.float32(dxdy[0])), float(np.float32(dxdy[1]))
        return t

    def ray_points_cone(self, dx, num_neg, num_pos, rays_neg, rays_pos, angle):
        n = (num_neg + num_pos + 1) * (rays_neg + 1 + rays_pos)
        out = np.empty((n, 2), np.float32)
        self.lib.orc_ray_points_cone(C.c_float(dx), C.c_int(num_neg), C.c_int(num_pos), C.c_int(rays_neg),
                                     C.c_int(rays_pos), C.c_float(angle), _fp(out))
        return out

    def terrain_grid_index(self, t, xy):
        xy = f32(xy); n = xy.shape[0]
        out = np.empty((n, 2), np.int64)
        self.lib.orc_terrain_grid_index(C.byref(t), _fp(xy), _ip64(out), C.c_int(n))
        return out

    def terrain_hf_vals(self, t, xy):
        xy = f32(xy); n = xy.shape[0]
        out = np.empty(n, np.float32)
        self.lib.orc_terrain_hf_vals(C.byref(t), _fp(xy), _fp(out), C.c_int(n))
        return out

    # ---- env ----------------------------------------------------------------------------
    @staticmethod
    def make_cfg(num_envs, key_body_ids, tar_obs_steps, ray_points, control_freq, episode_length, min_obs_h,
                 max_obs_h, reward_w, joint_err_w, dof_err_w, contact_weights, pose_termination_dist,
                 root_pos_termination_dist, root_rot_termination_angle, env_offsets, motion_offsets,
                 terrains_per_motion=1, enable_early_termination=True, pose_termination=True, track_root=True,
                 track_root_h=True, ema_weight=0.01, contact_body_ids=(), termination_height=0.15, global_obs=False, global_root_height_obs=False,
                 use_contact_info=True, enable_tar_obs=True):
        cfg = OrcEnvCfg()
        keep = []

        def hold(a):
            a = f32(a); keep.append(a); return _fp(a)

        cfg.num_envs = num_envs
        cfg.num_key = len(key_body_ids)
        for i, b in enumerate(key_body_ids): cfg.key_body_ids[i] = int(b)
        cfg.num_tar_steps = len(tar_obs_steps)
        for i, s in enumerate(tar_obs_steps): cfg.tar_obs_steps[i] = int(s)
        rp = f32(ray_points)
        cfg.num_rays = rp.shape[0]; cfg.ray_points = hold(rp)
        cfg.timestep_d = 1.0 / control_freq; cfg.timestep = 1.0 / control_freq
        cfg.episode_length = episode_length; cfg.min_obs_h = min_obs_h; cfg.max_obs_h = max_obs_h
        tw = sum(reward_w)
        cfg.pose_w, cfg.vel_w, cfg.root_pos_w, cfg.root_vel_w, cfg.key_pos_w = [w / tw for w in reward_w]
        cfg.joint_err_w = hold(joint_err_w); cfg.dof_err_w = hold(dof_err_w)
        cfg.contact_weights = hold(contact_weights); cfg.pose_termination_dist = hold(pose_termination_dist)
        cfg.root_pos_termination_dist = root_pos_termination_dist
        cfg.root_rot_termination_angle = root_rot_termination_angle
        cfg.enable_early_termination = int(enable_early_termination); cfg.pose_termination = int(pose_termination)
        cfg.track_root = int(track_root); cfg.track_root_h = int(track_root_h)
        cfg.ema_weight = ema_weight
        cfg.env_offsets = hold(env_offsets); cfg.motion_offsets = hold(motion_offsets)
        cfg.terrains_per_motion = terrains_per_motion
        cfg.num_contact_bodies = len(contact_body_ids)   # contact_bodies != []: fall termination (mgdm_dm_util.py:349-360)
        for i, b in enumerate(contact_body_ids): cfg.contact_body_ids[i] = int(b)
        cfg.termination_height = termination_height
        cfg.global_obs = int(global_obs)
        cfg.root_height_obs = int(global_root_height_obs)
        cfg.no_contact_info = int(not use_contact_info)
        cfg.no_tar_obs = int(not enable_tar_obs)
        cfg._keep = keep
        return cfg

    @staticmethod
    def make_state(num_envs, B=15, D=28, R=441, obs_w=1312, M=1, tracking_error=True):
        J = B - 1
        n = num_envs
        a = dict(
            char_root_pos=np.zeros((n, 3), np.float32), char_root_rot=np.zeros((n, 4), np.float32),
            char_root_vel=np.zeros((n, 3), np.float32), char_root_ang_vel=np.zeros((n, 3), np.float32),
            char_dof_pos=np.zeros((n, D), np.float32), char_dof_vel=np.zeros((n, D), np.float32),
            char_body_pos=np.zeros((n, B, 3), np.float32), contact_forces=np.zeros((n, B, 3), np.float32),
            motion_ids=np.zeros(n, np.int64), terrain_ids=np.zeros(n, np.int64),
            time_offsets=np.zeros(n, np.float32), time_buf=np.zeros(n, np.float32),
            timestep_buf=np.zeros(n, np.int32),
            ref_root_pos=np.zeros((n, 3), np.float32), ref_root_rot=np.zeros((n, 4), np.float32),
            ref_root_vel=np.zeros((n, 3), np.float32), ref_root_ang_vel=np.zeros((n, 3), np.float32),
            ref_joint_rot=np.zeros((n, J, 4), np.float32), ref_dof_pos=np.zeros((n, D), np.float32),
            ref_dof_vel=np.zeros((n, D), np.float32), ref_body_pos=np.zeros((n, B, 3), np.float32),
            ref_contacts=np.zeros((n, B), np.float32),
            ray_hfs=np.zeros((n, R), np.float32), obs=np.zeros((n, obs_w), np.float32),
            reward=np.zeros(n, np.float32), reward_terms=np.zeros((7, n), np.float32),
            tracking_error=np.zeros((n, 7), np.float32) if tracking_error else None,
            done=np.zeros(n, np.int32), fail_rates=np.ones(M, np.float32))
        return a

    @staticmethod
    def state_struct(a):
        s = OrcEnvState()
        for name, _t in OrcEnvState._fields_:
            arr = a.get(name)
            if arr is None:
                continue
            if arr.dtype == np.float32: setattr(s, name, _fp(arr))
            elif arr.dtype == np.int64: setattr(s, name, _ip64(arr))
            elif arr.dtype == np.int32: setattr(s, name, _ip32(arr))
        return s

    def env_post_physics_step(self, c, lib, t, cfg, state, begin=0, end=None):
        s = self.state_struct(state)
        end = cfg.num_envs if end is None else end
        self.lib.orc_env_post_physics_step(C.byref(c), lib, C.byref(t), C.byref(cfg), C.byref(s), C.c_int(begin), C.c_int(end))

    def env_update_curriculum(self, lib, cfg, state):
        s = self.state_struct(state)
        self.lib.orc_env_update_curriculum(lib, C.byref(cfg), C.byref(s))

    def env_reset_with(self, c, lib, t, cfg, state, env_ids, motion_ids, terrain_ids, t0, xy_noise):
        s = self.state_struct(state)
        env_ids = np.ascontiguousarray(env_ids, np.int64); motion_ids = np.ascontiguousarray(motion_ids, np.int64)
        terrain_ids = np.ascontiguousarray(terrain_ids, np.int64); t0 = f32(t0); xy_noise = f32(xy_noise)
        self.lib.orc_env_reset_with(C.byref(c), lib, C.byref(t), C.byref(cfg), C.byref(s), _ip64(env_ids),
                                    C.c_int(len(env_ids)), _ip64(motion_ids), _ip64(terrain_ids), _fp(t0), _fp(xy_noise))

    def env_refresh_rays(self, t, cfg, state, begin=0, end=None):
        s = self.state_struct(state)
        end = cfg.num_envs if end is None else end
        self.lib.orc_env_refresh_rays(C.byref(t), C.byref(cfg), C.byref(s), C.c_int(begin), C.c_int(end))

    def env_compute_obs(self, c, lib, cfg, state, env_ids):
        s = self.state_struct(state)
        env_ids = np.ascontiguousarray(env_ids, np.int64)
        self.lib.orc_env_compute_obs(C.byref(c), lib, C.byref(cfg), C.byref(s), _ip64(env_ids), C.c_int(len(env_ids)))

    def compute_done(self, cfg, B, time, root_rot, body_pos, tar_root_rot, tar_body_pos):
        time = f32(time); n = len(time)
        out = np.empty(n, np.int32)
        self.lib.orc_compute_done(C.byref(cfg), C.c_int(B), _fp(time), _fp(f32(root_rot)), _fp(f32(body_pos)),
                                  _fp(f32(tar_root_rot)), _fp(f32(tar_body_pos)), _ip32(out), C.c_int(n))
        return out

    def contact_reward(self, tar, forces, w):
        tar = f32(tar); forces = f32(forces); w = f32(w); n, B = tar.shape
        out = np.empty((n, B), np.float32)
        self.lib.orc_contact_reward(_fp(tar), _fp(forces), _fp(w), _fp(out), C.c_int(n), C.c_int(B))
        return out
