"""Episode statistics per reward key (mirror of ``return_tracker.py:6-90``): running means of the per-episode sums of
every entry of ``info["rewards"]`` (``total_r`` is the return) and of the episode length."""
import torch

from parc_amd.envs import base_env


class ReturnTracker:
    def __init__(self, num_envs, device, reward_keys):
        self._device = device
        self._episodes = 0
        self._mean_ep_len = torch.zeros([1], device=device, dtype=torch.float32)
        self._ep_len_buf = torch.zeros([num_envs], device=device, dtype=torch.long)
        self._eps_per_env_buf = torch.zeros([num_envs], device=device, dtype=torch.long)
        self._return_bufs = {k: torch.zeros([num_envs], device=device, dtype=torch.float32) for k in reward_keys}
        self._mean_returns = {k: torch.zeros([1], device=device, dtype=torch.float32) for k in reward_keys}

    def get_mean_return(self, key="total_r"):
        return self._mean_returns[key]

    def get_all_mean_returns(self):
        return {"mean_" + k: v.item() for k, v in self._mean_returns.items()}

    def get_specific_mean_return(self, key):
        return self._mean_returns[key]

    def get_mean_ep_len(self):
        return self._mean_ep_len

    def get_episodes(self):
        return self._episodes

    def get_eps_per_env(self):
        return self._eps_per_env_buf

    def reset(self):
        self._episodes = 0
        self._eps_per_env_buf[:] = 0
        self._ep_len_buf[:] = 0
        self._mean_ep_len = torch.zeros_like(self._mean_ep_len)
        for k in self._mean_returns:
            self._mean_returns[k] = torch.zeros_like(self._mean_returns[k])
            self._return_bufs[k][:] = 0.0

    def update(self, info, done):
        rewards = info["rewards"]
        for k in self._return_bufs:
            assert k in rewards, f"Missing reward key: {k}"
            self._return_bufs[k] += rewards[k]
        self._ep_len_buf += 1
        ids = (done != base_env.DoneFlags.NULL.value).nonzero(as_tuple=False).flatten()
        n = ids.shape[0]
        if n > 0:
            new_count = self._episodes + n
            w_new, w_old = float(n) / new_count, float(self._episodes) / new_count
            self._mean_ep_len = w_new * torch.mean(self._ep_len_buf[ids].float()) + w_old * self._mean_ep_len
            self._episodes = new_count
            for k in self._return_bufs:
                self._mean_returns[k] = w_new * torch.mean(self._return_bufs[k][ids]) + w_old * self._mean_returns[k]
                self._return_bufs[k][ids] = 0.0
            self._ep_len_buf[ids] = 0
            self._eps_per_env_buf[ids] += 1
