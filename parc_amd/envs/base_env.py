"""Gym-style base class: the contract the learner drives (mirror of ``PARC/motion_tracker/envs/base_env.py:9-72``)."""
import abc
import enum

import numpy as np


class EnvMode(enum.Enum):
    TRAIN = 0
    TEST = 1


class DoneFlags(enum.Enum):
    NULL = 0
    FAIL = 1
    SUCC = 2
    TIME = 3


class Box:
    """The slice of ``gym.spaces.Box`` the agent reads: ``shape, dtype, low, high`` (base_agent.py:163-219)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low); high = np.asarray(high)
            shape = low.shape
            dtype = low.dtype if np.issubdtype(low.dtype, np.floating) else dtype
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def __repr__(self):
        return f"Box(shape={self.shape}, dtype={self.dtype})"


class BaseEnv(abc.ABC):
    def __init__(self, visualize):
        self._mode = EnvMode.TRAIN
        self._visualize = visualize
        self._action_space = None

    @abc.abstractmethod
    def reset(self, env_ids=None):
        return

    @abc.abstractmethod
    def step(self, action):
        return

    def get_obs_space(self):
        obs, _ = self.reset()
        return Box(low=-np.inf, high=np.inf, shape=list(obs.shape[1:]), dtype=np.float32)

    def get_action_space(self):
        return self._action_space

    def set_mode(self, mode):
        self._mode = mode

    def get_num_envs(self):
        return int(1)

    def get_reward_bounds(self):
        return (-np.inf, np.inf)

    def get_reward_fail(self):
        return 0.0

    def get_reward_succ(self):
        return 0.0

    def get_visualize(self):
        return self._visualize

    def get_extra_log_info(self):
        return

    def post_test_update(self):
        return
