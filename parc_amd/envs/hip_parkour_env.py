"""``HipParkourEnv``: the reference's ``IGParkourEnv`` surface on top of ``libparc_env.so``.

Same constructor signature, same ``reset(env_ids) / step(action)`` contract, same ``info`` keys and the
private attributes the learner / recorder touch (``ig_parkour_env.py``, ``dm_env.py``,
``ig_env.py``).  PyTorch owns every per-env tensor (so the agent can read them with zero copies); the library
gets raw device pointers and enqueues its kernels on ``torch.cuda.current_stream()``.  There is no CPU
path: constructing the env without a GPU or without the built library raises.
"""
from __future__ import annotations

import ctypes as C
import os
import time
import types
from collections import OrderedDict

import numpy as np
import torch

from parc_amd import lib as L
from parc_amd import ms_file, terrain as terrain_mod
from parc_amd.envs import base_env, scene as scene_mod
from parc_amd.motion_lib import LoopMode


def _device_index(device) -> int:
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"HipParkourEnv needs a ROCm GPU device (got {device!r}); there is no CPU fallback")
    return 0 if d.index is None else d.index


def _quat_to_exp_map_np(q):
    """torch_util.py:372 (quat_to_axis_angle :70-91, then axis * angle) on a [n, 4] xyzw array, fp32 like the reference."""
    q = np.asarray(q, np.float32).copy()
    q[q[:, 3] < 0] *= np.float32(-1.0)
    ln = np.linalg.norm(q[:, :3], axis=-1).astype(np.float32)
    angle = (np.float32(2.0) * np.arctan2(ln, q[:, 3])).astype(np.float32)
    axis = q[:, :3] / np.maximum(ln, np.float32(1e-6))[:, None]
    small = ln <= np.float32(1e-5)
    angle[small] = 0.0
    axis[small] = np.array([0.0, 0.0, 1.0], np.float32)
    return (axis * angle[:, None]).astype(np.float32)


class _DMView:
    """What the learner / recorder reach through ``env.get_dm_env()`` (``dm_env.py``)."""

    def __init__(self, env):
        self._env = env
        self._ema_weight = 0.01

    @property
    def _motion_id_fail_rates(self):  # dm_ppo_agent.py:360-362 saves this tensor
        return self._env.get_fail_rates()

    @property
    def _motion_ids(self):
        return self._env._motion_ids

    @property
    def _terrains_per_motion(self):
        return self._env._scene.grid.terrains_per_motion

    def set_motion_start_time_fraction(self, val):  # dm_env.py:746-748
        self._env.set_reset_motion_start_time_fraction(val)

    def set_demo_mode(self, val=None):
        return self._env.set_demo_mode(val)

    def get_env_motion_length(self, env_ids):
        return self._env._motion_lengths[self._env._motion_ids[env_ids].long()]

    def get_env_motion_time(self, env_ids):
        return self._env._time_buf[env_ids] + self._env._motion_time_offsets[env_ids]

    def get_env_motion_name(self, env_id):
        return self._env._motion_names[int(self._env._motion_ids[env_id].item())]

    def get_extra_log_info(self):
        return self._env._dm_extra_log_info()

    def post_test_update(self):
        return


class HipParkourEnv(base_env.BaseEnv):
    NAME = "hip_parkour"

    def __init__(self, config, num_envs, device, visualize=False, env_id_base=0, total_envs=None, seed=0,
                 mirror_ref_state=True, enable_dynamics=None, dev_options=None, env_offsets=None):
        super().__init__(visualize=False)
        self._start_compute_time = time.time()
        self._lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipParkourEnv: no ROCm GPU visible (torch.cuda.is_available() is False)")
        self._config = config
        self._device = device
        self._num_envs = num_envs
        self._num_dm_envs = num_envs
        env_config = config["env"]
        self._env_config = env_config
        self._scene = sc = scene_mod.build_scene(config, num_envs, _device_index(device), env_id_base, total_envs,
                                                 seed=seed, enable_dynamics=enable_dynamics, dev_options=dev_options, env_offsets=env_offsets)
        self._kin_char_model = sc.char_model
        self._episode_length_val = env_config["episode_length"]
        self._control_freq = env_config["control_freq"]
        self._timestep = 1.0 / self._control_freq
        self._report_tracking_error = bool(env_config.get("report_tracking_error", False))
        self._output_motion_dir = env_config.get("output_motion_dir", "output/_motions/recorded_motions/")
        # "ms_file": the motion-terrain container MotionLib loads (default); "legacy_dict": the dict the reference's recorder pickles
        # (ig_parkour_env.py:698-736: frames / contacts / obs / obs_shapes / terrain)
        self._record_format = env_config.get("record_format", "ms_file")
        if self._record_format not in ("ms_file", "legacy_dict"):
            raise ValueError("record_format must be 'ms_file' or 'legacy_dict'")
        self._rand_reset = env_config.get("rand_reset", True)
        self._demo_mode = env_config["demo_mode"]
        self._rand_root_pos_offset_scale = env_config["rand_root_pos_offset_scale"]
        self._never_done = env_config.get("never_done", False)
        self._write_agent_states_flag = False
        self._bypass_record_fail = False
        self._record_ref = False
        self._record_obs = False
        self._rec = None
        self._motion_names = [c.name for c in sc.clips]
        self._key_body_ids = torch.tensor(sc.key_body_ids, dtype=torch.long, device=device)
        self._tar_obs_steps = torch.tensor(env_config.get("tar_obs_steps", [1]), dtype=torch.int, device=device)

        # ---- library handle + tables ----------------------------------------------------------------
        self._handle = C.c_void_p()
        L.check(self._lib.parc_env_create(C.byref(sc.cfg), C.byref(self._handle)))
        hw = self._lib.parc_env_health_words(self._handle)   # {flag-wait timeouts, manifold drops}: host-mapped, refreshed by every step
        self._health = np.ctypeslib.as_array(hw, shape=(2,)) if hw else None
        p = sc.packed
        mc = L.ParcMotionClips()
        mc.num_motions = len(sc.clips)
        mc.num_frames_host = L.np_i32p(p["num_frames"]); mc.fps_host = L.np_i32p(p["fps"])
        mc.loop_modes_host = L.np_i32p(p["loop_modes"]); mc.weights_host = p["weights"].ctypes.data_as(L.f64p)
        mc.root_pos_host = L.np_f32p(p["root_pos"]); mc.root_rot_host = L.np_f32p(p["root_rot"])
        mc.joint_rot_host = L.np_f32p(p["joint_rot"]); mc.contacts_host = L.np_f32p(p["contacts"])
        L.check(self._lib.parc_env_load_motions(self._handle, C.byref(mc)))
        g = sc.grid
        hf = np.ascontiguousarray(g.terrain.hf, np.float32)
        mo = np.ascontiguousarray(g.motion_offsets, np.float32)
        if mo.shape[0] != len(sc.clips):
            raise ValueError("terrain cache does not match the motion list (delete terrain_save_path)")
        L.check(self._lib.parc_env_load_terrain(self._handle, L.np_f32p(hf), hf.shape[0], hf.shape[1],
                                                float(g.terrain.min_point[0]), float(g.terrain.min_point[1]),
                                                float(g.terrain.dxdy[0]), float(g.terrain.dxdy[1]), L.np_f32p(mo),
                                                mo.shape[0], mo.shape[1]))
        M = len(sc.clips)
        lengths = np.zeros(M, np.float32); weights = np.zeros(M, np.float32)
        L.check(self._lib.parc_env_get_motion_info(self._handle, L.np_f32p(lengths), L.np_f32p(weights), M))
        self._motion_lengths = torch.from_numpy(lengths).to(device)
        self._motion_weights = torch.from_numpy(weights).to(device)
        self._num_motions = M
        self._obs_dim = self._lib.parc_env_obs_dim(self._handle)

        self._build_buffers(mirror_ref_state)
        low, high = sc.action_low, sc.action_high
        self._action_space = base_env.Box(low=low, high=high)
        self._action_bound_low = torch.tensor(low, device=device, dtype=torch.float32)
        self._action_bound_high = torch.tensor(high, device=device, dtype=torch.float32)
        if "fail_rates_path" in env_config["dm"]:  # dm_env.py:84-86
            print("LOADING SAVED FAIL RATES")
            fr = torch.load(env_config["dm"]["fail_rates_path"], weights_only=True).to(dtype=torch.float32).cpu().numpy()
            self.set_fail_rates(fr)
        if self._never_done:  # ig_parkour_env.py:980: the kernel writes NULL flags and an empty reset list
            L.check(self._lib.parc_env_set_never_done(self._handle, 1))
        self._dm_view = _DMView(self)
        self._info = dict()
        self.set_write_agent_states_flag(env_config.get("write_agent_states", False))
        if self._demo_mode:
            print("DEMO MODE ENABLED")

    # ------------------------------------------------------------------------------------------------
    def _build_buffers(self, mirror_ref_state):
        n, dev = self._num_envs, self._device
        cm = self._kin_char_model
        B, D, J = cm.get_num_bodies(), cm.get_dof_size(), cm.get_num_bodies() - 1
        R = self._scene.ray_points.shape[0]
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
        self._char_root_pos = z(n, 3); self._char_root_rot = z(n, 4); self._char_root_rot[:, 3] = 1.0
        self._char_root_vel = z(n, 3); self._char_root_ang_vel = z(n, 3)
        self._char_dof_pos = z(n, D); self._char_dof_vel = z(n, D)
        self._char_rigid_body_pos = z(n, B, 3)
        self._char_contact_forces = z(n, B, 3)
        self._motion_ids = z(n, dtype=torch.int32); self._motion_terrain_ids = z(n, dtype=torch.int32)
        self._motion_time_offsets = z(n)
        self._timestep_buf = z(n, dtype=torch.int32); self._time_buf = z(n)
        self._ep_num_buf = z(n, dtype=torch.int64)
        self._obs_buf = z(n, self._obs_dim); self._reward_buf = z(n); self._done_buf = z(n, dtype=torch.int32)
        self._reward_terms = z(7, n)
        self._tracking_error = z(n, 7) if self._report_tracking_error else None
        self._motion_start_time_fraction = z(n)
        ref = {}
        if mirror_ref_state:
            ref = dict(ref_root_pos=z(n, 3), ref_root_rot=z(n, 4), ref_root_vel=z(n, 3), ref_root_ang_vel=z(n, 3),
                       ref_joint_rot=z(n, J, 4), ref_dof_pos=z(n, D), ref_dof_vel=z(n, D), ref_body_pos=z(n, B, 3),
                       ref_contacts=z(n, B), ray_hfs=z(n, R))
        for k, v in ref.items():
            setattr(self, "_" + k, v)
        self._mirror_ref_state = mirror_ref_state
        b = L.ParcEnvBuffers()
        names = dict(char_root_pos=self._char_root_pos, char_root_rot=self._char_root_rot, char_root_vel=self._char_root_vel,
                     char_root_ang_vel=self._char_root_ang_vel, char_dof_pos=self._char_dof_pos,
                     char_dof_vel=self._char_dof_vel, char_body_pos=self._char_rigid_body_pos,
                     contact_forces=self._char_contact_forces, motion_ids=self._motion_ids,
                     terrain_ids=self._motion_terrain_ids, time_offsets=self._motion_time_offsets,
                     timestep=self._timestep_buf, time=self._time_buf, ep_num=self._ep_num_buf, obs=self._obs_buf,
                     reward=self._reward_buf, done=self._done_buf, reward_terms=self._reward_terms,
                     tracking_error=self._tracking_error, **ref)
        ptr_t = {"f": L.f32p, "i": L.i32p, "l": L.i64p}
        for name, kind in L.BUFFER_FIELDS:
            t = names.get(name)
            if t is not None:
                assert t.is_contiguous()
                setattr(b, name, C.cast(t.data_ptr(), ptr_t[kind]))
        self._bufs = b
        L.check(self._lib.parc_env_bind_buffers(self._handle, C.byref(b)))
        L.check(self._lib.parc_env_set_start_time_fraction(self._handle, self._motion_start_time_fraction.data_ptr()))

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.parc_env_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- BaseEnv contract ---------------------------------------------------------------------------
    def get_num_envs(self):
        return self._num_envs

    def get_reward_bounds(self):  # ig_char_env.py:42
        return (0.0, 1.0)

    def get_obs_space(self):  # ig_env.py:86-96
        return base_env.Box(low=-np.inf, high=np.inf, shape=[self._obs_dim], dtype=np.float32)

    def reset(self, env_ids=None):
        """ig_parkour_env.py:809-829.  ``None`` = all envs; a (possibly empty) LongTensor = that subset."""
        if env_ids is None:
            L.check(self._lib.parc_env_reset(self._handle, None, -1, self._stream()))
        else:
            env_ids = env_ids.to(device=self._device, dtype=torch.long).contiguous()
            if env_ids.numel() > 0:
                L.check(self._lib.parc_env_reset(self._handle, env_ids.data_ptr(), int(env_ids.numel()), self._stream()))
        self._update_info()
        return self._obs_buf, self._info

    def reset_done(self):
        """``BaseAgent._reset_done_envs`` without the ``nonzero()`` host sync: every env whose done flag was raised by
        the last ``step`` is re-sampled and reset on the device (``parc_env_reset_done``)."""
        L.check(self._lib.parc_env_reset_done(self._handle, self._stream()))
        return self._obs_buf, self._info

    def reset_with(self, env_ids, motion_ids, terrain_ids, t0, xy_noise):
        """Reset with the random draws injected (parity tests; ``parc_env_reset_with``)."""
        dev = self._device
        env_ids = env_ids.to(device=dev, dtype=torch.long).contiguous()
        mids = motion_ids.to(device=dev, dtype=torch.int32).contiguous()
        tids = terrain_ids.to(device=dev, dtype=torch.int32).contiguous()
        t0 = t0.to(device=dev, dtype=torch.float32).contiguous()
        nz = xy_noise.to(device=dev, dtype=torch.float32).contiguous()
        L.check(self._lib.parc_env_reset_with(self._handle, env_ids.data_ptr(), int(env_ids.numel()), mids.data_ptr(),
                                              tids.data_ptr(), t0.data_ptr(), nz.data_ptr(), self._stream()))
        self._update_info()
        return self._obs_buf, self._info

    def step(self, action):
        """ig_env.py:66-84: returns the persistent (obs, reward, done, info) tensors."""
        a = None
        if action is not None:
            assert action.shape == (self._num_envs, self._char_dof_pos.shape[1]) and action.dtype == torch.float32
            a = action.contiguous().data_ptr()
        L.check(self._lib.parc_env_step(self._handle, a, self._stream()))
        if self._health is not None and self._health[0] != 0:   # host-mapped counter of the dynamics kernel, refreshed by every step: no sync
            self.check_health()
        self.write_agent_states()  # ig_parkour_env.py:686-696: after update_done, before the agent resets anything
        self._update_info()
        return self._obs_buf, self._reward_buf, self._done_buf, self._info

    def step_and_reset_done(self, action):
        """``step(action)`` followed by ``reset_done()`` as ONE hipGraph launch (launch-bound sizes: a few thousand envs).
        Returns the same persistent tensors; the rows of finished envs already hold the first observation of their new
        episode, ``done`` / ``reward`` still describe the step that ended."""
        if self._rec is not None and self.is_writing_agent_states():
            raise RuntimeError("the recorder needs step() + reset_done(); the graph step does not record")
        if action is not None:
            if getattr(self, "_action_buf", None) is None:
                self._action_buf = torch.zeros_like(self._char_dof_pos)
                L.check(self._lib.parc_env_bind_action(self._handle, self._action_buf.data_ptr()))
            if action.data_ptr() != self._action_buf.data_ptr():
                self._action_buf.copy_(action)
        L.check(self._lib.parc_env_step_reset_graph(self._handle, self._stream()))
        if self._health is not None and self._health[0] != 0:
            self.check_health()
        self._update_info()
        return self._obs_buf, self._reward_buf, self._done_buf, self._info

    def _update_info(self):
        """ig_parkour_env.py:1108-1116 + the reward dict of :1012-1044 (views, not clones: the agent copies them)."""
        rt = self._reward_terms
        self._info["rewards"] = {"pose_r": rt[0], "vel_r": rt[1], "root_pos_r": rt[2], "root_vel_r": rt[3],
                                 "key_pos_r": rt[4], "contact_penalty": rt[5], "total_r": rt[6]}
        self._info["timestep"] = self._timestep_buf
        self._info["ep_num"] = self._ep_num_buf
        self._info["compute_time"] = time.time() - self._start_compute_time
        self._info["char_contact_forces"] = self._char_contact_forces
        if self._report_tracking_error:
            self._info["tracking_error"] = self._tracking_error

    def _update_reward(self):
        """The agent calls this once to size its return tracker (base_agent.py:229-234)."""
        self._update_info()

    def _compute_obs(self, env_ids=None, ret_obs_shapes=False):
        if ret_obs_shapes:
            print("OBS SHAPES")
            for k, v in self._scene.obs_shapes.items():
                print(k, v)
            return self._scene.obs_shapes
        if env_ids is None:
            L.check(self._lib.parc_env_compute_obs(self._handle, None, -1, self._stream()))
            return self._obs_buf
        env_ids = env_ids.to(device=self._device, dtype=torch.long).contiguous()
        L.check(self._lib.parc_env_compute_obs(self._handle, env_ids.data_ptr(), int(env_ids.numel()), self._stream()))
        return self._obs_buf[env_ids]

    # ---- dm env surface ------------------------------------------------------------------------------
    def has_dm_envs(self):
        return True

    def get_dm_env(self):
        return self._dm_view

    def get_fail_rates(self):
        out = np.zeros(self._num_motions, np.float32)
        L.check(self._lib.parc_env_get_fail_rates(self._handle, L.np_f32p(out), self._num_motions))
        return torch.from_numpy(out)

    def set_fail_rates(self, fr):
        fr = np.ascontiguousarray(np.asarray(fr, np.float32))
        L.check(self._lib.parc_env_set_fail_rates(self._handle, L.np_f32p(fr), self._num_motions))

    def _push_reset_mode(self):
        L.check(self._lib.parc_env_set_rand_reset(self._handle, int(bool(self._rand_reset)), int(bool(self._demo_mode)),
                                                  float(self._rand_root_pos_offset_scale)))

    def set_rand_reset(self, val=None):
        self._rand_reset = (not self._rand_reset) if val is None else val
        self._push_reset_mode()
        print("Setting rand reset to:", self._rand_reset)

    def set_demo_mode(self, val=None):
        self._demo_mode = (not self._demo_mode) if val is None else val
        self._push_reset_mode()
        return self._demo_mode

    def set_rand_root_pos_offset_scale(self, val):
        self._rand_root_pos_offset_scale = val
        self._push_reset_mode()

    def set_reset_motion_start_time_fraction(self, val):
        self._motion_start_time_fraction[:] = val.to(self._device)

    def set_output_motion_dir(self, path):
        self._output_motion_dir = path

    def set_write_agent_states_flag(self, val):
        self._write_agent_states_flag = val

    def is_writing_agent_states(self):
        return self._write_agent_states_flag

    # ---- recorder (ig_parkour_env.py:698-796, 1155-1181; driven by learning/dm_motion_recorder.py) ---------------
    @property
    def _episode_length(self):
        return self._episode_length_val

    @_episode_length.setter
    def _episode_length(self, val):
        self._episode_length_val = float(val)
        if getattr(self, "_handle", None):
            L.check(self._lib.parc_env_set_episode_length(self._handle, float(val)))

    def build_agent_states_dict(self, name_suffix="", record_obs=False, max_frames=None):
        """Start recording every env.  The reference keeps Python lists per env; here rows go to device ring buffers
        ``[cap][N][78]`` (+ obs ``[cap][N][1312]``), cap = the longest clip at the control rate + margin."""
        N, B, dev = self._num_envs, len(self._kin_char_model.get_body_names()), self._device
        if max_frames is None:
            max_frames = int(np.ceil(float(self._motion_lengths.max().item()) * self._control_freq)) + 8
        W = 3 + 4 + 4 * (B - 1) + B
        rec = types.SimpleNamespace()
        rec.cap = int(max_frames)
        rec.frames = torch.zeros(rec.cap, N, W, dtype=torch.float32, device=dev)
        rec.obs = torch.zeros(rec.cap, N, self._obs_dim, dtype=torch.float32, device=dev) if record_obs else None
        rec.count = torch.zeros(N, dtype=torch.int32, device=dev)
        rec.writing = torch.ones(N, dtype=torch.uint8, device=dev)
        rec.n_writing = torch.full((1,), N, dtype=torch.int32, device=dev)
        rec.host_writing = np.ones(N, bool)
        rec.obs_shapes = self._compute_obs(None, ret_obs_shapes=True) if record_obs else None
        self._rec = rec
        self._record_obs = record_obs
        L.check(self._lib.parc_env_record_bind(self._handle, rec.frames.data_ptr(), rec.obs.data_ptr() if record_obs else None, rec.cap,
                                               rec.count.data_ptr(), rec.writing.data_ptr(), rec.n_writing.data_ptr(),
                                               int(bool(self._record_ref))))
        os.makedirs(self._output_motion_dir, exist_ok=True)  # ig_parkour_env.py:56-58 does this when the env is built
        self.set_write_agent_states_flag(True)
        self._env_success_state = [False] * N
        self._save_motion_name_suffix = name_suffix
        print("recording agent motion..")

    def is_writing_env_state(self, env_id):
        return bool(self._rec.host_writing[env_id])

    def set_writing_env_state(self, env_id, val):
        """Only switching an env OFF before the roll-out starts is meaningful (dm_motion_recorder.py:66-68)."""
        rec = self._rec
        if bool(val) != bool(rec.host_writing[env_id]):
            rec.host_writing[env_id] = bool(val)
            rec.writing[env_id] = 1 if val else 0
            rec.n_writing += 1 if val else -1

    def set_env_success_state(self, env_id, val):
        self._env_success_state[env_id] = val

    def get_env_success_states(self):
        return self._env_success_state

    def write_agent_states(self):
        """Append the current state of every recording env (one kernel), then save the envs that just finished."""
        if not self.is_writing_agent_states() or self._rec is None:
            return
        rec = self._rec
        L.check(self._lib.parc_env_record_frame(self._handle, self._stream()))
        now = rec.writing.cpu().numpy().astype(bool)        # the recorder is host-driven like the reference's; one small D2H per step
        finished = np.nonzero(rec.host_writing & ~now)[0]
        rec.host_writing = now
        self.set_write_agent_states_flag(bool(now.any()))
        for env_id in finished:
            env_id = int(env_id)
            dm = self.get_dm_env()
            motion_length = dm.get_env_motion_length(env_id).item()
            curr_motion_time = dm.get_env_motion_time(env_id).item()
            motion_name = dm.get_env_motion_name(env_id)
            if not self._bypass_record_fail and curr_motion_time < motion_length - self._timestep * 2.0:
                print("env", env_id, "failed to track motion", motion_name)
                continue
            self.set_env_success_state(env_id, True)
            self.save_agent_states_to_file(env_id, motion_name + self._save_motion_name_suffix)

    def save_agent_states_to_file(self, env_id, output_motion_name=None):
        """ig_parkour_env.py:698-736: global xy, terrain sliced around the trajectory and localised, one file per env —
        written in the motion-terrain container of file_io.py (what MotionLib loads), obs in ``misc_data``."""
        rec = self._rec
        n = int(rec.count[env_id].item())
        rows = rec.frames[:n, env_id].cpu().numpy()
        B = len(self._kin_char_model.get_body_names())
        root_pos = rows[:, 0:3].copy()
        root_pos[:, 0:2] += self._scene.env_offsets[env_id, 0:2]  # _get_global_xy_pos
        gt = self._scene.grid.terrain
        padding = round(1.0 // float(gt.dxdy[0])) * float(gt.dxdy[0])
        sliced, local = terrain_mod.slice_terrain_around_motion(root_pos, gt, padding=padding)
        md = ms_file.MSMotionData(root_pos=local.astype(np.float32), root_rot=rows[:, 3:7].copy(),
                                  joint_rot=rows[:, 7:7 + 4 * (B - 1)].reshape(n, B - 1, 4).copy(),
                                  body_contacts=rows[:, 7 + 4 * (B - 1):].copy(), fps=int(self._control_freq), loop_mode="CLAMP")
        misc = None
        if self._record_obs:
            shapes = {k: {"use_normalizer": bool(v["use_normalizer"]), "shape": tuple(v["shape"])} for k, v in rec.obs_shapes.items()}
            misc = {"obs": rec.obs[:n, env_id].cpu().numpy(), "obs_shapes": shapes, "hf_mask_inds": None}  # plain data only
        if output_motion_name is None:
            output_motion_name = "dm_motion_" + str(env_id).zfill(3)
        os.makedirs(self._output_motion_dir, exist_ok=True)
        path = os.path.join(self._output_motion_dir, output_motion_name + ".pkl")
        if self._record_format == "legacy_dict":
            self._save_legacy_dict(path, rec, env_id, n, local, sliced, md.body_contacts)
            return
        ms_file.save_ms_file(ms_file.MSFileData(motion_data=md, terrain_data=sliced.to_ms_terrain_data(), misc_data=misc), path)
        print("wrote motion data to", path)
        print("num frames =", n)

    def _save_legacy_dict(self, path, rec, env_id, n, local_root_pos, sliced_terrain, contacts):
        """The reference recorder's own output (ig_parkour_env.py:698-736, rows of ``_get_char_state`` :664-685): one pickled dict
        {fps, loop_mode, frames [n, 3 + 3 + D] = localised root position | root exponential map | dofs, contacts [n, B],
        obs, obs_shapes, terrain}.  The reference pickles its SubTerrain object under ``terrain``; here it is the same fields as
        plain arrays (a file must not need this package's classes to load).  Root exp map / dofs are recovered from the recorded
        quaternions with the reference's conversions (torch_util.py:70-91,372; kin_char_model.py:601 on the device)."""
        import pickle
        B = len(self._kin_char_model.get_body_names())
        D = self._char_dof_pos.shape[1]
        rows = rec.frames[:n, env_id].contiguous()
        jr = rows[:, 7:7 + 4 * (B - 1)].contiguous()
        dof = torch.zeros(n, D, dtype=torch.float32, device=self._device)
        if n > 0:
            L.check(self._lib.parc_rot_to_dof(self._handle, jr.data_ptr(), dof.data_ptr(), n, self._stream()))
        frames = np.concatenate([np.asarray(local_root_pos, np.float32), _quat_to_exp_map_np(rows[:, 3:7].cpu().numpy()),
                                 dof.cpu().numpy()], axis=1).astype(np.float32)
        out = {"fps": int(self._control_freq), "loop_mode": "CLAMP", "frames": frames, "contacts": np.asarray(contacts, np.float32)}
        if self._record_obs:
            out["obs"] = rec.obs[:n, env_id].cpu().numpy()
            out["obs_shapes"] = {k: {"use_normalizer": bool(v["use_normalizer"]), "shape": tuple(v["shape"])} for k, v in rec.obs_shapes.items()}
        t = sliced_terrain
        out["terrain"] = {"hf": np.asarray(t.hf, np.float32), "hf_maxmin": np.asarray(t.hf_maxmin, np.float32), "min_point": np.asarray(t.min_point, np.float32),
                          "dxdy": np.asarray(t.dxdy, np.float32), "dims": np.asarray(t.hf.shape, np.int64)}
        with open(path, "wb") as f:
            pickle.dump(out, f)
        print("wrote motion data to", path)
        print("num frames =", n)

    def _dm_extra_log_info(self):
        """dm_env.py:668-727 (per-motion fail rates + quantiles)."""
        fr = self.get_fail_rates()
        names = self._motion_names
        top, ids = torch.sort(fr, descending=True)
        q = torch.tensor(self._env_config["dm"]["fail_rate_quantiles"], dtype=torch.float32)
        at_q = torch.quantile(top, q)
        info = {"MOTION_FAIL_RATES": {names[i]: fr[i].item() * 100.0 for i in range(len(names))},
                "Misc": {"top fail rate": top[0].item() * 100.0}}
        for i in range(q.shape[0]):
            info["Misc"]["Fail Rate at " + str(round(q[i].item() * 100.0)) + "% Quantile"] = at_q[i].item() * 100.0
        return info

    def describe(self):
        """What the library handle resolved to (dynamics kernel, block size, collision set, contact parameters, manifold period, curriculum
        path, the developer switches it was created with) as a dict: bench.py records it with every measurement."""
        txt = self._lib.parc_env_describe(self._handle).decode()
        out = {}
        for kv in txt.split(";"):
            if "=" in kv:
                k, v = kv.split("=", 1)
                out[k] = v
        i = txt.find("dev_options=")
        if i >= 0:
            out["dev_options"] = txt[i + len("dev_options="):]
        return out

    def dynamics_timeouts(self):
        """Flag waits of the dynamics kernel that hit their bound since the library was loaded (must be 0; synchronises the device).
        A block that saw one wrote NaN root positions for its envs (parc_dynamics_wave.hpp)."""
        return int(self._lib.parc_env_dynamics_timeouts(self._handle))

    def dynamics_manifold_drops(self):
        """Contact planes the dynamics kernel had no room for since the library was loaded (its per-lane plane list: LDS share + overflow
        area; parc_dynamics_wave.hpp).  A dropped plane is one contact of a body that already has twenty others: not an error, but it
        should not happen in practice -- the GPU tests assert 0 on the benchmark scenes (synchronises the device)."""
        return int(self._lib.parc_env_dynamics_manifold_drops(self._handle))

    def check_health(self):
        """Raise when the dynamics kernel reported a hand-off timeout: the physics of this run cannot be trusted."""
        n = self.dynamics_timeouts()
        if n > 0:
            raise L.ParcError(f"k_dynamics_wave: {n} LDS-flag hand-off(s) timed out; the affected envs hold NaN root positions. "
                              "This is a bug in the library (or a broken build), not a property of the scene: stop the run.")
        return n

    def get_extra_log_info(self):
        """dm_env.py:668-727, plus the health counter of the dynamics kernel (read here = every iters_per_output iterations:
        the query synchronises).  A non-zero counter aborts the run (scripts/run_tracker.py exits non-zero)."""
        info = dict(self._dm_extra_log_info())
        info["Env_Health"] = {"Dynamics_Flag_Timeouts": float(self.check_health()),
                              "Dynamics_Manifold_Drops": float(self.dynamics_manifold_drops())}
        return info

    def post_test_update(self):
        return

    # ---- measurement ------------------------------------------------------------------------------------
    def set_kernel_timing(self, enable):
        """hipEvents around the kernels of every step() from now on (no host sync); read with get_kernel_timing()."""
        L.check(self._lib.parc_env_set_kernel_timing(self._handle, int(bool(enable))))

    def get_kernel_timing(self):
        """(dynamics ms, obs ms, steps) averaged over the steps since the last call; plus the HBM figures of the obs kernels
        (HBM_GBps, Roofline_Frac) from the algorithmic 5 772 B per env-step and the 8 TB/s peak."""
        d, o, n = C.c_double(), C.c_double(), C.c_int32()
        L.check(self._lib.parc_env_get_kernel_timing(self._handle, C.byref(d), C.byref(o), C.byref(n)))
        out = {"dynamics_ms": d.value, "obs_ms": o.value, "steps": n.value}
        if o.value > 0.0:
            out["hbm_gbps"] = 5772.0 * self._num_envs / (o.value * 1e-3) / 1e9
            out["roofline_frac"] = out["hbm_gbps"] / 8000.0
        return out

    def profile_step(self, iters=10, action=None):
        a = None if action is None else action.contiguous().data_ptr()
        tot = C.c_float(); post = C.c_float()
        L.check(self._lib.parc_env_profile_step(self._handle, a, self._stream(), iters, C.byref(tot), C.byref(post)))
        return tot.value, post.value
