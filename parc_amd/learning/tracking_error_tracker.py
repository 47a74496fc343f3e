"""Per-episode means of the 7 tracking-error terms the env reports (``tracking_error_tracker.py:6-125``; the terms
themselves come from the step kernel: ``mgdm_dm_util.compute_tracking_error:521-553``).

Column order of ``info["tracking_error"]``: root_pos, root_rot, body_pos, body_rot, dof_vel, root_vel, root_ang_vel.
Same update rule as the reference (sum per env, divide by the episode length at the end of an episode, running mean over
finished episodes weighted by episode count) on ``[7]`` / ``[N,7]`` tensors instead of seven scalars."""
import torch

from parc_amd.envs import base_env

NAMES = ("root_pos", "root_rot", "body_pos", "body_rot", "dof_vel", "root_vel", "root_ang_vel")


class TrackingErrorTracker:
    def __init__(self, num_envs, device):
        self._device = device
        self._episodes = 0
        self._mean = torch.zeros(7, device=device, dtype=torch.float32)
        self._err_buf = torch.zeros(num_envs, 7, device=device, dtype=torch.float32)
        self._ep_len_buf = torch.zeros(num_envs, device=device, dtype=torch.long)

    def reset(self):
        self._episodes = 0
        self._mean.zero_()
        self._err_buf.zero_()
        self._ep_len_buf.zero_()

    def get_episodes(self):
        return self._episodes

    def get_mean(self, name):
        return self._mean[NAMES.index(name)]

    def get_mean_root_pos_err(self): return self._mean[0]
    def get_mean_root_rot_err(self): return self._mean[1]
    def get_mean_body_pos_err(self): return self._mean[2]
    def get_mean_body_rot_err(self): return self._mean[3]
    def get_mean_dof_vel_err(self): return self._mean[4]
    def get_mean_root_vel_err(self): return self._mean[5]
    def get_mean_root_ang_vel_err(self): return self._mean[6]

    def update(self, tracking_error, done):
        assert tracking_error.shape == self._err_buf.shape and done.shape[0] == self._err_buf.shape[0]
        self._err_buf += tracking_error
        self._ep_len_buf += 1
        reset_ids = (done != base_env.DoneFlags.NULL.value).nonzero(as_tuple=False).flatten()
        num_resets = int(reset_ids.numel())
        if num_resets > 0:
            new_count = self._episodes + num_resets
            w_new, w_old = float(num_resets) / new_count, float(self._episodes) / new_count
            per_ts = self._err_buf[reset_ids] / self._ep_len_buf[reset_ids].unsqueeze(-1)
            self._mean = w_new * per_ts.mean(dim=0) + w_old * self._mean
            self._episodes += num_resets
            self._err_buf[reset_ids] = 0.0
            self._ep_len_buf[reset_ids] = 0

    def test_info(self):
        """Keys the reference adds to the test info (dm_ppo_agent.py:164-180)."""
        return {"test_mean_root_pos_tracking_err": self._mean[0].item(), "test_mean_root_rot_tracking_err": self._mean[1].item(),
                "test_mean_body_pos_tracking_err": self._mean[2].item(), "test_mean_body_rot_tracking_err": self._mean[3].item(),
                "test_mean_dof_vel_tracking_err": self._mean[4].item(), "test_mean_root_vel_tracking_err": self._mean[5].item(),
                "test_mean_root_ang_vel_tracking_err": self._mean[6].item()}
