"""PPO learner for the tracker (``agent_name: DM_PPO``).

One class with the surface of the reference's ``BaseAgent`` → ``PPOAgent`` → ``DMPPOAgent`` chain
(``base_agent.py``, ``ppo_agent.py``, ``dm_ppo_agent.py``): same config keys, same method names, same state-dict
keys (``_obs_norm._mean``, ``_a_norm._std``, ``_model._actor_layers.0.weight`` …) so reference checkpoints load,
same ``model.pt`` / ``checkpoints/model_%010d.pt`` / ``fail_rates_%010d.pt`` outputs and log keys.

New relative to the reference:
  * data parallel over env shards: one process per GPU; every optimizer step all-reduces one flat gradient
    bucket over RCCL (``optimizer.py``); normalizer sums and advantage statistics are all-reduced once per
    iteration; the curriculum fail-rate table is averaged across ranks once per iteration;
  * ``env.reset_done()`` (device-side reset of finished envs) replaces the ``nonzero()`` host round trip when the
    env provides it.
"""
import os
import time

import numpy as np
import torch

from parc_amd.envs import base_env
from parc_amd.learning import (dist_util, experience_buffer, normalizer, optimizer, ppo_model, return_tracker, rl_util,
                               tracking_error_tracker)
from parc_amd.util.logger import Logger


class AgentMode:
    TRAIN = 0
    TEST = 1


class DMPPOAgent(torch.nn.Module):
    NAME = "DM_PPO"

    def __init__(self, config, env, device):
        super().__init__()
        self._env = env
        self._device = device
        self._iter = 0
        self._sample_count = 0
        self._config = config
        self._load_params(config)
        self._build_normalizers()
        self._model = ppo_model.DMPPOModel(config["model"], env)
        self.to(device)
        params = [p for p in self.parameters() if p.requires_grad]
        self._optimizer = optimizer.Optimizer(config["optimizer"], params)
        self._build_exp_buffer(config)
        self._env._update_reward()
        keys = list(self._env._info["rewards"].keys())
        self._train_return_tracker = return_tracker.ReturnTracker(self.get_num_envs(), device, keys)
        self._test_return_tracker = return_tracker.ReturnTracker(self.get_num_envs(), device, keys)
        self._test_tracking_error_tracker = None
        if getattr(env, "_report_tracking_error", False):  # dm_ppo_agent.py:25-28
            self._test_tracking_error_tracker = tracking_error_tracker.TrackingErrorTracker(self.get_num_envs(), device)
        self._mode = AgentMode.TRAIN
        self._curr_obs = None
        self._curr_info = None
        self._logger = None

    # ---- config (base_agent.py:147-157, ppo_agent.py:23-52) -------------------------------------------------
    def _load_params(self, config):
        self._discount = config["discount"]
        self._iters_per_output = config["iters_per_output"]
        self._iters_per_checkpoint = config["iters_per_checkpoint"]
        self._normalizer_samples = config.get("normalizer_samples", np.inf)
        self._test_episodes = config["test_episodes"]
        self._steps_per_iter = config["steps_per_iter"]
        self._update_epochs = config["update_epochs"]
        self._batch_size = config["batch_size"]
        self._td_lambda = config["td_lambda"]
        self._ppo_clip_ratio = config["ppo_clip_ratio"]
        self._norm_adv_clip = config["norm_adv_clip"]
        self._action_bound_weight = config["action_bound_weight"]
        self._action_entropy_weight = config["action_entropy_weight"]
        self._action_reg_weight = config["action_reg_weight"]
        self._critic_loss_weight = config["critic_loss_weight"]
        self._exp_anneal_samples = config.get("exp_anneal_samples", np.inf)
        self._exp_prob_beg = config.get("exp_prob_beg", 1.0)
        self._exp_prob_end = config.get("exp_prob_end", 1.0)
        self._clip_grad_norm = config.get("clip_grad_norm", False)
        self._max_grad_norm = config.get("max_grad_norm", 0.5)
        self._critic_loss_type = config.get("critic_loss_type", "L2")
        self._use_reset_done = bool(config.get("device_side_reset", True))

    def _build_normalizers(self):
        """dm_ppo_agent.py:48-86: blocks flagged ``use_normalizer: False`` keep mean 0 / std 1."""
        obs_space = self._env.get_obs_space()
        shapes = self._env._compute_obs(ret_obs_shapes=True)
        idx, cur = [], 0
        for key in shapes:
            shape = shapes[key]["shape"]
            flat = shape[0] * shape[1] if len(shape) >= 2 else shape[0]
            if not shapes[key]["use_normalizer"]:
                idx.append(torch.arange(cur, cur + flat, dtype=torch.int64, device=self._device))
            cur += flat
        non_norm = torch.cat(idx, dim=0) if idx else None
        self._obs_norm = normalizer.Normalizer(tuple(obs_space.shape), device=self._device, dtype=torch.float32,
                                               non_norm_indices=non_norm, clip=self._config["norm_obs_clip"])
        a_space = self._env.get_action_space()
        a_mean = torch.tensor(0.5 * (a_space.high + a_space.low), device=self._device, dtype=torch.float32)
        a_std = torch.tensor(0.5 * (a_space.high - a_space.low), device=self._device, dtype=torch.float32)
        self._a_norm = normalizer.Normalizer(tuple(a_mean.shape), device=self._device, init_mean=a_mean, init_std=a_std)

    def _build_exp_buffer(self, config):
        T, N, dev = self._steps_per_iter, self.get_num_envs(), self._device
        self._exp_buffer = experience_buffer.ExperienceBuffer(buffer_length=T, batch_size=N, device=dev)
        obs_dim = list(self._env.get_obs_space().shape)
        a_dim = list(self._env.get_action_space().shape) or [1]
        z = lambda *s, dtype=torch.float: torch.zeros([T, N] + list(s), device=dev, dtype=dtype)
        for name, buf in [("obs", z(*obs_dim)), ("next_obs", z(*obs_dim)), ("action", z(*a_dim)), ("reward", z()),
                          ("done", z(dtype=torch.int)), ("a_logp", z()), ("tar_val", z()), ("adv", z()),
                          ("rand_action_mask", z())]:
            self._exp_buffer.add_buffer(name, buf)

    # ---- small accessors ----------------------------------------------------------------------------------
    def get_num_envs(self):
        return self._env.get_num_envs()

    def get_env(self):
        return self._env

    def get_action_size(self):
        return int(np.prod(self._env.get_action_space().shape))

    def calc_num_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def set_mode(self, mode):
        self._mode = mode
        self._env.set_mode(base_env.EnvMode.TRAIN if mode == AgentMode.TRAIN else base_env.EnvMode.TEST)

    def eval_mode(self):
        self.eval()
        self.set_mode(AgentMode.TEST)

    # ---- checkpoints (base_agent.py:128-140, dm_ppo_agent.py:357-362) ----------------------------------------
    def save(self, out_file):
        d = os.path.dirname(out_file)
        if d:
            os.makedirs(d, exist_ok=True)
        torch.save(self.state_dict(), out_file)

    def load(self, in_file):
        state_dict = torch.load(in_file, map_location=self._device, weights_only=True)
        self.load_state_dict(state_dict)
        self._optimizer.sync()
        Logger.print("Loaded model parameters from {:s}".format(str(in_file)))

    def _output_train_model(self, it, out_model_file, int_output_dir):
        if dist_util.rank() != 0:
            return
        self.save(out_model_file)
        if int_output_dir != "":
            self.save(os.path.join(int_output_dir, "model_{:010d}.pt".format(it)))
            if self._env.has_dm_envs():
                torch.save(self._env.get_dm_env()._motion_id_fail_rates.cpu(), os.path.join(int_output_dir, "fail_rates_{:010d}.pt".format(it)))

    # ---- training loop (dm_ppo_agent.py:191-233) -------------------------------------------------------------
    def train_model(self, max_samples, out_model_file, int_output_dir, log_file, logger_type=None):
        start_time = time.time()
        self._curr_obs, self._curr_info = self._env.reset()
        self._logger = Logger()
        self._logger.set_step_key("Samples")
        self._logger.configure_output_file(log_file if dist_util.rank() == 0 else None)
        self._init_train()
        test_info = {"mean_return": 0.0, "mean_ep_len": 0.0, "num_eps": 0}
        while self._sample_count < max_samples:
            train_info = self._train_iter()
            output_iter = (self._iter % self._iters_per_output == 0)
            if output_iter:
                test_info = self.test_model(self._test_episodes)
                extra = self._env.get_extra_log_info() or {}
                for collection in extra:
                    for k, v in extra[collection].items():
                        self._logger.log(k, v, collection=collection, quiet=True)
                self._env.post_test_update()
            self._sample_count = self._exp_buffer.get_total_samples() * dist_util.world_size()
            self._log_train_info(train_info, test_info, start_time)
            if dist_util.rank() == 0:
                self._logger.print_log()
            if output_iter:
                if dist_util.rank() == 0:
                    self._logger.write_log()
                self._train_return_tracker.reset()
                self.hard_reset_envs()
            if self._iter % self._iters_per_checkpoint == 0:
                self._output_train_model(self._iter, out_model_file, int_output_dir)
            self._iter += 1

    def _init_train(self):
        self._iter = 0
        self._sample_count = 0
        self._exp_buffer.clear()
        self._train_return_tracker.reset()
        self._test_return_tracker.reset()

    def hard_reset_envs(self):
        self._curr_obs, self._curr_info = self._env.reset()

    def test_model(self, num_episodes):
        self.eval()
        self.set_mode(AgentMode.TEST)
        self.hard_reset_envs()
        return self._rollout_test(num_episodes)

    def _train_iter(self):
        self._exp_buffer.reset()
        self.eval()
        self.set_mode(AgentMode.TRAIN)
        timed = hasattr(self._env, "set_kernel_timing")
        if timed:
            self._env.set_kernel_timing(True)
        self._rollout_train(self._steps_per_iter)
        kt = None
        if timed:
            kt = self._env.get_kernel_timing()
            self._env.set_kernel_timing(False)
        data_info = self._build_train_data()
        train_info = self._update_model()
        if self._need_normalizer_update():
            self._obs_norm.update()
        self._merge_fail_rates()
        info = {**train_info, **data_info}
        info["mean_return"] = self._train_return_tracker.get_mean_return().item()
        info["mean_ep_len"] = self._train_return_tracker.get_mean_ep_len().item()
        info["num_eps"] = self._train_return_tracker.get_episodes()
        info.update(self._train_return_tracker.get_all_mean_returns())
        if kt is not None and kt["steps"] > 0:  # env kernel time and HBM figures next to the learning curves
            info["env_dynamics_ms"] = kt["dynamics_ms"]; info["env_obs_ms"] = kt["obs_ms"]
            info["hbm_gbps"] = kt.get("hbm_gbps", 0.0); info["roofline_frac"] = kt.get("roofline_frac", 0.0)
        return info

    def _merge_fail_rates(self):
        """Curriculum table averaged over ranks (each rank's EMA only sees its own shard)."""
        if dist_util.is_dist() and self._env.has_dm_envs():
            fr = self._env.get_dm_env()._motion_id_fail_rates.to(self._device, dtype=torch.float32)
            dist_util.all_reduce_mean_(fr)
            self._env.set_fail_rates(fr.cpu().numpy())

    # ---- rollout (base_agent.py:291-370) --------------------------------------------------------------------
    def _rollout_train(self, num_steps):
        obs_buf = self._exp_buffer.get_data("obs")
        for _ in range(num_steps):
            # normalise for the policy and write the raw row into the rollout buffer in one pass over the observation
            norm_obs = self._obs_norm.normalize_and_record(self._curr_obs, obs_buf[self._exp_buffer.get_buffer_head()])
            action, action_info = self._decide_action(self._curr_obs, self._curr_info, norm_obs=norm_obs)
            self._record_data_pre_step(self._curr_obs, self._curr_info, action, action_info, obs_recorded=True)
            next_obs, r, done, next_info = self._step_env(action)
            self._train_return_tracker.update(next_info, done)
            self._record_data_post_step(next_obs, r, done, next_info)
            self._curr_obs, self._curr_info = self._reset_done_envs(done)
            self._exp_buffer.inc()

    def _rollout_test(self, num_episodes):
        self._test_return_tracker.reset()
        tet = self._test_tracking_error_tracker
        if tet is not None:
            tet.reset()
        if num_episodes == 0:
            return {"mean_return": 0.0, "mean_ep_len": 0.0, "num_eps": 0}
        min_eps_per_env = int(np.ceil(num_episodes / self.get_num_envs()))
        while True:
            action, _ = self._decide_action(self._curr_obs, self._curr_info)
            next_obs, r, done, next_info = self._step_env(action)
            self._test_return_tracker.update(next_info, done)
            if tet is not None and "tracking_error" in next_info:
                tet.update(next_info["tracking_error"], done)
            self._curr_obs, self._curr_info = self._reset_done_envs(done)
            if torch.all(self._test_return_tracker.get_eps_per_env() > min_eps_per_env - 1):
                break
        info = {"mean_return": self._test_return_tracker.get_mean_return().item(),
                "mean_ep_len": self._test_return_tracker.get_mean_ep_len().item(),
                "num_eps": self._test_return_tracker.get_episodes()}
        info.update(self._test_return_tracker.get_all_mean_returns())
        if tet is not None:
            info.update(tet.test_info())
        return info

    def step(self):
        action, action_info = self._decide_action(self._curr_obs, self._curr_info)
        next_obs, r, done, next_info = self._step_env(action)
        return next_obs, r, done, next_info, action, action_info

    def reset(self):
        self._curr_obs, self._curr_info = self._env.reset()

    def _step_env(self, action):
        return self._env.step(action)

    def _reset_done_envs(self, done):
        if self._use_reset_done and hasattr(self._env, "reset_done"):
            return self._env.reset_done()
        ids = torch.flatten((done != base_env.DoneFlags.NULL.value).nonzero(as_tuple=False))
        return self._env.reset(ids)

    def _need_normalizer_update(self):
        return self._sample_count < self._normalizer_samples

    # ---- acting (ppo_agent.py:84-122) -------------------------------------------------------------------------
    @torch.no_grad()
    def _decide_action(self, obs, info, norm_obs=None):
        if norm_obs is None:
            norm_obs = self._obs_norm.normalize(obs)
        dist = self._model.eval_actor(norm_obs)
        if self._mode == AgentMode.TRAIN:
            norm_a_rand, norm_a_mode = dist.sample(), dist.mode
            exp_prob = torch.full([norm_a_rand.shape[0], 1], self._get_exp_prob(), device=self._device, dtype=torch.float)
            mask = torch.bernoulli(exp_prob)
            norm_a = torch.where(mask == 1.0, norm_a_rand, norm_a_mode)
            mask = mask.squeeze(-1)
        else:
            norm_a = dist.mode
            mask = torch.zeros_like(norm_a[..., 0])
        logp = dist.log_prob(norm_a).detach()
        a = self._a_norm.unnormalize(norm_a.detach())
        return a, {"a_logp": logp, "rand_action_mask": mask}

    def _get_exp_prob(self):
        if np.isfinite(self._exp_anneal_samples):
            l = float(np.clip(float(self._sample_count) / self._exp_anneal_samples, 0.0, 1.0))
            return (1.0 - l) * self._exp_prob_beg + l * self._exp_prob_end
        return self._exp_prob_beg

    def _record_data_pre_step(self, obs, info, action, action_info, obs_recorded=False):
        if not obs_recorded:
            self._exp_buffer.record("obs", obs)
        self._exp_buffer.record("action", action)
        if self._need_normalizer_update():
            self._obs_norm.record(obs)
        self._exp_buffer.record("a_logp", action_info["a_logp"])
        self._exp_buffer.record("rand_action_mask", action_info["rand_action_mask"])

    def _record_data_post_step(self, next_obs, r, done, next_info):
        self._exp_buffer.record("next_obs", next_obs)
        self._exp_buffer.record("reward", r)
        self._exp_buffer.record("done", done)

    # ---- targets (ppo_agent.py:124-171) ----------------------------------------------------------------------
    @torch.no_grad()
    def _build_train_data(self):
        self.eval()
        buf = self._exp_buffer
        obs, next_obs = buf.get_data("obs"), buf.get_data("next_obs")
        r, done, mask = buf.get_data("reward"), buf.get_data("done"), buf.get_data("rand_action_mask")
        next_vals = self._eval_critic_chunked(next_obs)
        r_min, r_max = self._env.get_reward_bounds()
        next_vals = torch.clamp(next_vals, r_min / (1.0 - self._discount), r_max / (1.0 - self._discount))
        next_vals[done == base_env.DoneFlags.SUCC.value] = self._env.get_reward_succ() / (1.0 - self._discount)
        next_vals[done == base_env.DoneFlags.FAIL.value] = self._env.get_reward_fail() / (1.0 - self._discount)
        new_vals = rl_util.compute_td_lambda_return(r, next_vals, done, self._discount, self._td_lambda)
        vals = self._eval_critic_chunked(obs)
        adv = new_vals - vals
        sel = adv.flatten()[(mask == 1.0).flatten()]
        # advantage statistics over ALL ranks' samples (unbiased std like torch.std_mean)
        stats = torch.stack([sel.sum().double(), torch.square(sel).sum().double(), torch.tensor(float(sel.numel()), device=sel.device, dtype=torch.float64)])
        dist_util.all_reduce_sum_(stats)
        n = stats[2].clamp_min(2.0)
        adv_mean = (stats[0] / n).float()
        adv_std = torch.sqrt(torch.clamp_min((stats[1] - n * torch.square(stats[0] / n)) / (n - 1.0), 0.0)).float()
        norm_adv = torch.clamp((adv - adv_mean) / torch.clamp_min(adv_std, 1e-5), -self._norm_adv_clip, self._norm_adv_clip)
        buf.set_data("tar_val", new_vals)
        buf.set_data("adv", norm_adv)
        return {"adv_mean": adv_mean, "adv_std": adv_std}

    def _eval_critic_chunked(self, obs):
        """Critic over [T, N, obs] one rollout step at a time (bounded activation memory at 65 536 envs)."""
        out = torch.empty(obs.shape[:2], device=obs.device, dtype=torch.float32)
        for t in range(obs.shape[0]):
            out[t] = self._model.eval_critic(self._obs_norm.normalize(obs[t])).squeeze(-1)
        return out

    # ---- update (ppo_agent.py:183-330) -------------------------------------------------------------------------
    def _update_model(self):
        self.train()
        num_envs = self.get_num_envs()
        num_samples = self._exp_buffer.get_sample_count()
        batch_size = self._batch_size * num_envs
        num_batches = int(np.ceil(float(num_samples) / batch_size))
        train_info = dict()
        for _ in range(self._update_epochs):
            for _ in range(num_batches):
                batch = self._exp_buffer.sample(batch_size)
                loss_info = self._compute_loss(batch)
                if self._clip_grad_norm:
                    self._optimizer.step(loss_info["loss"], model=self._model, max_norm=self._max_grad_norm)
                else:
                    self._optimizer.step(loss_info["loss"])
                for k, v in loss_info.items():
                    v = v.detach()
                    train_info[k] = train_info[k] + v if k in train_info else v
        steps = self._update_epochs * num_batches
        return {k: v / steps for k, v in train_info.items()}

    def _compute_loss(self, batch):
        batch["norm_obs"] = self._obs_norm.normalize(batch["obs"])
        batch["norm_action"] = self._a_norm.normalize(batch["action"])
        critic_info = self._compute_critic_loss(batch)
        actor_info = self._compute_actor_loss(batch)
        critic_loss, actor_loss = critic_info["critic_loss"], actor_info["actor_loss"]
        if critic_loss.item() > 20.0:  # ppo_agent.py:222-235: do not trust the critic's gradients for the actor
            print("LARGE CRITIC LOSS")
            actor_loss = actor_loss.detach()
        if torch.isnan(critic_loss).any() or torch.isnan(actor_loss).any():
            print("NAN LOSS\ncritic loss:", critic_loss, "\nactor loss:", actor_loss)
            os.makedirs("output", exist_ok=True)
            torch.save({k: v.detach().cpu() for k, v in batch.items()}, "output/debug_batch.pt")
            raise SystemExit("NaN loss: wrote output/debug_batch.pt")
        loss = actor_loss + self._critic_loss_weight * critic_loss
        return {"loss": loss, **critic_info, **actor_info}

    def _compute_critic_loss(self, batch):
        pred = self._model.eval_critic(batch["norm_obs"]).squeeze(-1)
        diff = batch["tar_val"] - pred
        loss = torch.mean(torch.square(diff)) if self._critic_loss_type == "L2" else torch.mean(torch.abs(diff))
        return {"critic_loss": loss}

    def _compute_actor_loss(self, batch):
        m = batch["rand_action_mask"] == 1.0  # only exploratory samples carry a policy gradient
        norm_obs, norm_a, old_logp, adv = batch["norm_obs"][m], batch["norm_action"][m], batch["a_logp"][m], batch["adv"][m]
        a_dist = self._model.eval_actor(norm_obs)
        ratio = torch.exp(a_dist.log_prob(norm_a) - old_logp)
        l0 = adv * ratio
        l1 = adv * torch.clamp(ratio, 1.0 - self._ppo_clip_ratio, 1.0 + self._ppo_clip_ratio)
        actor_loss = -torch.mean(torch.minimum(l0, l1))
        info = {"actor_loss": actor_loss,
                "clip_frac": torch.mean((torch.abs(ratio - 1.0) > self._ppo_clip_ratio).float()).detach(),
                "imp_ratio": torch.mean(ratio).detach()}
        if self._action_bound_weight != 0:  # base_agent.py:431-452: normalised actions should stay in [-1, 1]
            lo = torch.clamp_max(a_dist.mode + 1.0, 0.0)
            hi = torch.clamp_min(a_dist.mode - 1.0, 0.0)
            bound = torch.mean(torch.sum(torch.square(lo), dim=-1) + torch.sum(torch.square(hi), dim=-1))
            actor_loss = actor_loss + self._action_bound_weight * bound
            info["action_bound_loss"] = bound.detach()
        if self._action_entropy_weight != 0:
            ent = torch.mean(a_dist.entropy())
            actor_loss = actor_loss - self._action_entropy_weight * ent
            info["action_entropy"] = ent.detach()
        if self._action_reg_weight != 0:
            reg = torch.mean(a_dist.param_reg())
            actor_loss = actor_loss + self._action_reg_weight * reg
            info["action_reg_loss"] = reg.detach()
        info["actor_loss"] = actor_loss
        return info

    # ---- logging (base_agent.py:402-429) ----------------------------------------------------------------------
    def _log_train_info(self, train_info, test_info, start_time):
        lg = self._logger
        wall = (time.time() - start_time) / 3600.0
        lg.log("Iteration", self._iter, collection="1_Info")
        lg.log("Wall_Time", wall, collection="1_Info")
        lg.log("Samples", self._sample_count, collection="1_Info")
        lg.log("Test_Return", test_info["mean_return"], collection="0_Main")
        lg.log("Test_Episode_Length", test_info["mean_ep_len"], collection="0_Main")
        lg.log("Test_Episodes", test_info["num_eps"], collection="1_Info")
        train_info = dict(train_info)
        lg.log("Train_Return", train_info.pop("mean_return"), collection="0_Main")
        lg.log("Train_Episode_Length", train_info.pop("mean_ep_len"), collection="0_Main")
        lg.log("Train_Episodes", train_info.pop("num_eps"), collection="1_Info")
        for k, v in train_info.items():
            lg.log(k.title(), v.item() if torch.is_tensor(v) else v)
        lg.log("Exp_Prob", self._get_exp_prob())
        if wall > 0:
            lg.log("Env_Steps_Per_Sec", self._sample_count / (wall * 3600.0))
