"""Record the motions a trained tracker reproduces (PARC stage 4, ``learning/dm_motion_recorder.py:45-121`` driven by
``run_tracker.py --mode record``).

One env per motion (deterministic resets, demo mode); every env records its character until the clip ends, envs that
fall early are retried from later start times 10 %, 20 %, ... 50 % into the clip.  The per-step bookkeeping that the
reference does in Python lists lives in device ring buffers (``HipParkourEnv.build_agent_states_dict``); the files are
the motion-terrain containers of ``file_io.py`` and are binned into folders of 50 like the reference does."""
import copy
import os
import re
import shutil

import torch

from parc_amd.learning.dm_ppo_agent import AgentMode


def organize_recorded_dm_motions(source_dir, verbose=True):
    """dm_motion_recorder.py:13-43: ``<name>_<number>_*.pkl`` -> ``<name>_<lo>_<hi>/`` in bins of 50."""
    pattern = re.compile(r"^(.*?)_(\d+)_.*\.pkl$")
    if not os.path.isdir(source_dir):
        return
    for filename in os.listdir(source_dir):
        match = pattern.match(filename)
        if not match:
            continue
        base, number = match.group(1), int(match.group(2))
        lo = (number // 50) * 50
        dest = os.path.join(source_dir, f"{base}_{lo}_{lo + 49}")
        os.makedirs(dest, exist_ok=True)
        shutil.move(os.path.join(source_dir, filename), os.path.join(dest, filename))
        if verbose:
            print(f"Moved {filename} -> {os.path.basename(dest)}/")


def record_dm_motions(agent, start_time_fractions=(0.1, 0.2, 0.3, 0.4, 0.5), record_obs=True, organize=True, max_steps=None):
    """Returns (successful_motions [num_envs] bool list, num_successful per pass)."""
    agent.eval()
    agent.set_mode(AgentMode.TEST)
    env = agent.get_env()
    if not hasattr(env, "get_dm_env"):
        raise AttributeError("record_dm_motions requires an agent with a DM environment")
    env.set_rand_reset(False)
    env.set_demo_mode(True)
    env.set_rand_root_pos_offset_scale(0.0)
    env._episode_length = 1000.0

    def record_pass(name_suffix, prev_successful=None):
        agent._curr_obs, agent._curr_info = env.reset()
        env.build_agent_states_dict(name_suffix, record_obs=record_obs)
        if prev_successful is not None:  # the reference records frame 0 first and switches envs off afterwards; same files result
            for env_id, ok in enumerate(prev_successful):
                env.set_writing_env_state(env_id, not ok)
        env.write_agent_states()
        steps = 0
        while env.is_writing_agent_states():
            action, _ = agent._decide_action(agent._curr_obs, agent._curr_info)
            _, _, done, _ = agent._step_env(action)
            agent._curr_obs, agent._curr_info = agent._reset_done_envs(done)
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
        print("done writing agent states")

    record_pass("_dm")
    successful = copy.deepcopy(env.get_env_success_states())
    num_successful = [sum(successful)]
    num_envs = agent.get_num_envs()
    lengths = env.get_dm_env().get_env_motion_length(torch.arange(num_envs, device=agent._device))
    for frac in start_time_fractions:
        if all(successful):
            break
        for i in range(num_envs):  # clips with less than 2 s left are not retried (dm_motion_recorder.py:97-100)
            if (1.0 - frac) * lengths[i].item() < 2.0:
                successful[i] = True
        env.get_dm_env().set_motion_start_time_fraction(frac * torch.ones(num_envs, dtype=torch.float32, device=agent._device))
        record_pass("_dm", prev_successful=successful)
        new = copy.deepcopy(env.get_env_success_states())
        num_successful.append(sum(new))
        successful = [a or b for a, b in zip(successful, new)]
    if organize:
        print("Organizing motions...")
        organize_recorded_dm_motions(env._output_motion_dir, verbose=False)
        print("Finished organizing motions.")
    print("Successful motions at 0 percent start time:", num_successful[0])
    print("Success rate:", num_successful[0] / max(1, len(successful)))
    for i, n in enumerate(num_successful[1:]):
        print("Successful motions at", start_time_fractions[i], "percent start time:", n)
    print("Total successfull motions:", sum(num_successful))
    print("Total success rate:", sum(num_successful) / max(1, len(successful)))
    return successful, num_successful
