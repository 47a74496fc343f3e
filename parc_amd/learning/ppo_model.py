"""Actor / critic (mirror of ``ppo_model.py:8-57`` + ``base_model.py:9-28``): submodule names ``_actor_layers``,
``_action_dist``, ``_critic_layers``, ``_critic_out`` match the reference's state-dict keys."""
import numpy as np
import torch

from parc_amd.learning import distribution_gaussian_diag as dgd
from parc_amd.learning import nets


class PPOModel(torch.nn.Module):
    def __init__(self, config, env):
        super().__init__()
        self._activation = torch.nn.ReLU
        obs_space = env.get_obs_space()
        a_space = env.get_action_space()
        self._actor_layers, _ = nets.build_net(config["actor_net"], {"obs": obs_space}, self._activation)
        a_size = int(np.prod(a_space.shape))
        self._action_dist = dgd.DistributionGaussianDiagBuilder(
            nets.calc_layers_out_size(self._actor_layers), a_size, std_type=dgd.StdType[config["actor_std_type"]],
            init_std=config["action_std"], init_output_scale=config["actor_init_output_scale"])
        self._critic_layers, _ = nets.build_net(config["critic_net"], {"obs": obs_space}, self._activation)
        self._critic_out = torch.nn.Linear(nets.calc_layers_out_size(self._critic_layers), 1)
        torch.nn.init.zeros_(self._critic_out.bias)
        # Optional (not in the reference, off by default): run the two MLP trunks in bf16 on the matrix cores; the small heads
        # (action distribution, value) and everything downstream stay fp32.  `model: {amp: bf16}` in the agent config.
        amp = str(config.get("amp", "none")).lower()
        assert amp in ("none", "bf16"), amp
        self._amp_dtype = torch.bfloat16 if amp == "bf16" else None

    def _trunk(self, layers, obs):
        if self._amp_dtype is None or not obs.is_cuda:
            return layers(obs)
        with torch.autocast(device_type="cuda", dtype=self._amp_dtype):
            h = layers(obs)
        return h.float()

    def eval_actor(self, obs):
        return self._action_dist(self._trunk(self._actor_layers, obs))

    def eval_critic(self, obs):
        return self._critic_out(self._trunk(self._critic_layers, obs))


DMPPOModel = PPOModel  # dm_ppo_model.py:12 — only the default MLP branch is supported
