"""CPU tests of the PPO learner and its collectives (gloo, world_size 2)."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import DATA
from parc_amd.util import path_loader


def agent_config():
    cfg = copy.deepcopy(path_loader.load_config(os.path.join(DATA, "configs/tracker_config/dm_agent_default.yaml")))
    cfg["model"]["actor_net"] = "fc_2layers_128units"
    cfg["model"]["critic_net"] = "fc_2layers_128units"
    cfg["steps_per_iter"] = 8
    cfg["optimizer"]["learning_rate"] = 2e-3
    cfg["optimizer"]["type"] = "Adam"
    cfg["iters_per_output"] = 1000
    cfg["iters_per_checkpoint"] = 1000
    cfg["test_episodes"] = 0
    return cfg


def test_td_lambda_matches_bruteforce():
    from parc_amd.learning import rl_util
    g = torch.Generator().manual_seed(0)
    T, N = 9, 5
    r = torch.rand(T, N, generator=g); nv = torch.rand(T, N, generator=g)
    done = (torch.rand(T, N, generator=g) < 0.2).int()
    disc, lam = 0.99, 0.95
    out = rl_util.compute_td_lambda_return(r, nv, done, disc, lam)
    # brute force: lambda-weighted mix of n-step returns, truncated at episode ends / the buffer end
    ref = torch.zeros(T, N)
    for n in range(N):
        for t0 in range(T):
            ret, w_left, acc, g_ = 0.0, 1.0, 0.0, 1.0
            for t in range(t0, T):
                acc += g_ * r[t, n].item()
                nstep = acc + g_ * disc * nv[t, n].item()
                last = (t == T - 1) or done[t, n].item() != 0
                w = w_left if last else (1 - lam) * (lam ** (t - t0))
                ret += w * nstep
                w_left -= w
                g_ *= disc
                if last:
                    break
            ref[t0, n] = ret
    assert torch.allclose(out, ref, atol=1e-5)


def test_normalizer_matches_batch_statistics():
    from parc_amd.learning.normalizer import Normalizer
    nz = Normalizer((6,), "cpu", non_norm_indices=torch.tensor([4, 5]))
    x1, x2 = torch.randn(100, 6) * 3 + 1, torch.randn(50, 6) * 0.5 - 2
    nz.record(x1); nz.update(); nz.record(x2); nz.update()
    allx = torch.cat([x1, x2])
    assert torch.allclose(nz.get_mean()[:4], allx.mean(0)[:4], atol=1e-5)
    assert torch.allclose(nz.get_std()[:4], allx.std(0, unbiased=False)[:4], atol=1e-4)
    assert torch.all(nz.get_mean()[4:] == 0) and torch.all(nz.get_std()[4:] == 1)
    assert set(nz.state_dict().keys()) == {"_count", "_mean", "_std"}   # reference checkpoint keys


def test_state_dict_keys_are_reference_compatible():
    from fake_env import FakeEnv
    from parc_amd.learning.dm_ppo_agent import DMPPOAgent
    cfg = agent_config()
    cfg["model"]["actor_net"] = cfg["model"]["critic_net"] = "fc_3layers_2048units"
    agent = DMPPOAgent(cfg, FakeEnv(4, obs_dim=1312, act_dim=28), "cpu")
    keys = set(agent.state_dict().keys())
    for k in ["_obs_norm._mean", "_obs_norm._std", "_obs_norm._count", "_a_norm._mean", "_model._actor_layers.0.weight",
              "_model._actor_layers.4.bias", "_model._action_dist._mean_net.weight", "_model._action_dist._logstd_net",
              "_model._critic_layers.2.weight", "_model._critic_out.weight"]:
        assert k in keys, k
    assert agent.calc_num_params() == 10638877   # actor 5 326 364 + critic 5 312 513


def test_ppo_improves_return_single_process(tmp_path):
    from fake_env import FakeEnv
    from parc_amd.learning.dm_ppo_agent import DMPPOAgent
    torch.manual_seed(0)
    env = FakeEnv(64)
    agent = DMPPOAgent(agent_config(), env, "cpu")
    agent._curr_obs, agent._curr_info = env.reset()
    agent._logger = None
    agent._init_train()
    rets = []
    for it in range(40):
        info = agent._train_iter()
        agent._sample_count = agent._exp_buffer.get_total_samples()
        rets.append(agent._exp_buffer.get_data("reward").mean().item())
    assert np.mean(rets[-5:]) > np.mean(rets[:5]) + 0.01, (rets[:5], rets[-5:])
    agent.save(str(tmp_path / "m.pt"))
    agent2 = DMPPOAgent(agent_config(), FakeEnv(64), "cpu")
    agent2.load(str(tmp_path / "m.pt"))
    for a, b in zip(agent.state_dict().values(), agent2.state_dict().values()):
        assert torch.equal(a, b)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from fake_env import FakeEnv
    from parc_amd.learning.dm_ppo_agent import DMPPOAgent
    from parc_amd.learning import dist_util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # different init per rank: sync() must fix it
    env = FakeEnv(32, seed=rank)
    env.set_fail_rates(np.array([0.2, 0.4, 1.0]) if rank == 0 else np.array([0.6, 0.8, 1.0]))
    agent = DMPPOAgent(agent_config(), env, "cpu")
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    for _ in range(3):
        agent._train_iter()
        agent._sample_count = agent._exp_buffer.get_total_samples() * dist_util.world_size()
    flat = torch.cat([p.detach().flatten() for p in agent.parameters()])
    res = dict(params=flat, mean=agent._obs_norm.get_mean().clone(), count=agent._obs_norm.get_count().clone(),
               fail=env._fail.clone(), samples=agent._sample_count)
    torch.save(res, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo(tmp_path):
    """world_size 2 over gloo: gradients are all-reduced (identical parameters on both ranks), normalizer statistics
    and the fail-rate table are merged, samples are counted over both shards."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True); r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert torch.equal(r0["params"], r1["params"])
    assert torch.allclose(r0["mean"], r1["mean"]) and torch.equal(r0["count"], r1["count"])
    assert int(r0["count"].item()) == 3 * 8 * 32 * 2
    assert torch.allclose(r0["fail"], torch.tensor([0.4, 0.6, 1.0])) and torch.allclose(r1["fail"], r0["fail"])
    assert r0["samples"] == 3 * 8 * 32 * 2


def test_tracking_error_tracker_matches_reference_rule():
    """tracking_error_tracker.py:72-125: per-env sums / episode length, running mean over finished episodes."""
    import torch
    from parc_amd.learning.tracking_error_tracker import TrackingErrorTracker
    g = torch.Generator().manual_seed(0)
    n = 6
    trk = TrackingErrorTracker(n, "cpu")
    # straightforward restatement with python lists
    sums = [[0.0] * 7 for _ in range(n)]; lens = [0] * n; finished = []
    for step in range(40):
        te = torch.rand(n, 7, generator=g)
        done = (torch.rand(n, generator=g) < 0.15).int() * (1 + (step % 3))
        trk.update(te, done)
        for e in range(n):
            for k in range(7):
                sums[e][k] += te[e, k].item()
            lens[e] += 1
            if done[e] != 0:
                finished.append([s / lens[e] for s in sums[e]])
                sums[e] = [0.0] * 7; lens[e] = 0
    assert trk.get_episodes() == len(finished) > 5
    want = torch.tensor(finished).mean(0)
    assert torch.allclose(trk._mean, want, atol=1e-5)
    info = trk.test_info()
    assert abs(info["test_mean_dof_vel_tracking_err"] - want[4].item()) < 1e-5 and len(info) == 7


def test_td_lambda_against_brute_force_definition():
    """The reference keeps a brute-force TD(lambda) (the lambda-weighted mixture of n-step returns, truncated at episode
    ends) beside its recursion but never calls it (rl_util.py:32-74); restated here as the check it was meant to be."""
    import torch
    from parc_amd.learning import rl_util
    g = torch.Generator().manual_seed(4)
    T, N, disc, lam = 12, 9, 0.97, 0.9
    r = torch.rand(T, N, generator=g); nv = torch.randn(T, N, generator=g)
    done = (torch.rand(T, N, generator=g) < 0.2).int() * 2
    got = rl_util.compute_td_lambda_return(r, nv, done, disc, lam).numpy().astype(np.float64)
    rr, vv, dd = r.numpy().astype(np.float64), nv.numpy().astype(np.float64), done.numpy()
    want = np.zeros((T, N))
    for i in range(N):
        for t0 in range(T):
            new_val, sum_r, cd, cl = 0.0, 0.0, 1.0, 1.0
            for t in range(t0, T):
                sum_r += cd * rr[t, i]
                cur = sum_r + cd * disc * vv[t, i]
                if dd[t, i] == 0 and t < T - 1:
                    new_val += (1 - lam) * cl * cur
                else:
                    new_val += cl * cur
                    break
                cd *= disc; cl *= lam
            want[t0, i] = new_val
    assert np.abs(got - want).max() < 1e-5
