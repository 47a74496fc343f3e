"""TD(lambda) returns (``rl_util.py:7-30``): backward recursion over the rollout, lambda reset at episode ends."""
import torch

from parc_amd.envs import base_env


def compute_td_lambda_return(r, next_vals, done, discount, td_lambda):
    assert r.shape == next_vals.shape
    ret = torch.zeros_like(r)
    reset = (done != base_env.DoneFlags.NULL.value).type(torch.float)
    ret[-1] = r[-1] + discount * next_vals[-1]
    for i in reversed(range(r.shape[0] - 1)):
        lam = td_lambda * (1.0 - reset[i])
        ret[i] = r[i] + discount * ((1.0 - lam) * next_vals[i] + lam * ret[i + 1])
    return ret
