"""A tiny CPU env with the BaseEnv surface the learner uses — lets the PPO loop and its collectives run under
pytest without a GPU.  Reward = -|action - f(obs)|: learnable, so a few iterations must raise the return."""
import time
from collections import OrderedDict

import numpy as np
import torch

from parc_amd.envs import base_env


class FakeEnv(base_env.BaseEnv):
    def __init__(self, num_envs, device="cpu", obs_dim=24, act_dim=6, ep_len=16, seed=0):
        super().__init__(False)
        self._num_envs, self._device, self._obs_dim, self._act_dim, self._ep_len = num_envs, device, obs_dim, act_dim, ep_len
        self._config = {"env": {}}
        self._report_tracking_error = False
        self._g = torch.Generator(device="cpu").manual_seed(seed)
        self._W = torch.randn(obs_dim, act_dim, generator=torch.Generator().manual_seed(1234)) * 0.3
        self._obs_buf = torch.zeros(num_envs, obs_dim)
        self._reward_buf = torch.zeros(num_envs)
        self._done_buf = torch.zeros(num_envs, dtype=torch.int)
        self._t = torch.zeros(num_envs, dtype=torch.int)
        self._ep = torch.zeros(num_envs, dtype=torch.int64)
        self._action_space = base_env.Box(low=-np.ones(act_dim), high=np.ones(act_dim))
        self._info = {}
        self._fail = torch.ones(3)
        self._update_reward()

    def get_num_envs(self):
        return self._num_envs

    def get_obs_space(self):
        return base_env.Box(low=-np.inf, high=np.inf, shape=[self._obs_dim], dtype=np.float32)

    def get_reward_bounds(self):
        return (-10.0, 0.0)

    def _compute_obs(self, env_ids=None, ret_obs_shapes=False):
        if ret_obs_shapes:
            s = OrderedDict()
            s["char_obs"] = {"use_normalizer": True, "shape": (self._obs_dim - 4,)}
            s["hf"] = {"use_normalizer": False, "shape": (4,)}
            return s
        return self._obs_buf

    def _update_reward(self):
        self._info["rewards"] = {"total_r": self._reward_buf, "pose_r": self._reward_buf * 0.5}
        self._info["timestep"] = self._t
        self._info["ep_num"] = self._ep
        self._info["compute_time"] = time.time()
        self._info["char_contact_forces"] = torch.zeros(self._num_envs, 15, 3)

    def reset(self, env_ids=None):
        ids = torch.arange(self._num_envs) if env_ids is None else env_ids
        if len(ids) > 0:
            self._obs_buf[ids] = torch.randn(len(ids), self._obs_dim, generator=self._g)
            self._t[ids] = 0
            self._ep[ids] += 1
        return self._obs_buf, self._info

    def step(self, action):
        target = 0.5 * torch.tanh(self._obs_buf[:, :1] @ self._W[:1]) + 0.4   # mostly a constant offset: quick to learn
        self._reward_buf[:] = -torch.mean(torch.abs(action - target), dim=-1)
        self._t += 1
        self._done_buf[:] = torch.where(self._t >= self._ep_len, 3, 0).int()
        self._obs_buf[:] = 0.9 * self._obs_buf + 0.1 * torch.randn(self._num_envs, self._obs_dim, generator=self._g)
        self._update_reward()
        return self._obs_buf, self._reward_buf, self._done_buf, self._info

    def has_dm_envs(self):
        return True

    def get_dm_env(self):
        class _V:
            pass
        v = _V()
        v._motion_id_fail_rates = self._fail.clone()
        return v

    def set_fail_rates(self, fr):
        self._fail = torch.as_tensor(np.asarray(fr), dtype=torch.float32)

    def get_extra_log_info(self):
        return {"Misc": {"top fail rate": float(self._fail.max()) * 100.0}}
