#!/usr/bin/env python3
"""Train / test the tracker — same argv as the reference's ``scripts/run_tracker.py:96-134``:

    python scripts/run_tracker.py --mode train --num_envs 4096 --device cuda:0 --visualize false \\
        --env_config data/configs/tracker_config/dm_env_default.yaml \\
        --agent_config data/configs/tracker_config/dm_agent_default.yaml \\
        --out_model_file output/model.pt --int_output_dir output/checkpoints --log_file output/log.txt

Multi-GPU (new): launch with ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ...``;
every rank owns ``num_envs`` envs (global env index = rank * num_envs + i) and gradients are all-reduced over RCCL.
"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd.envs import env_builder  # noqa: E402
from parc_amd.learning import agent_builder, dist_util  # noqa: E402
from parc_amd.util import arg_parser, path_loader  # noqa: E402
from parc_amd.util.logger import Logger  # noqa: E402


def load_args(argv):
    args = arg_parser.ArgParser()
    args.load_args(argv[1:])
    arg_file = args.parse_string("arg_file", "")
    if arg_file != "":
        assert args.load_file(arg_file), "Failed to load args from: " + arg_file
    return args


def set_rand_seed(args):
    rand_seed = args.parse_int("rand_seed") if args.has_key("rand_seed") else int(np.uint64(time.time() * 256))
    rand_seed += dist_util.rank()
    print("Setting seed: {}".format(rand_seed))
    random.seed(rand_seed)
    np.random.seed(np.uint64(rand_seed % (2 ** 32)))
    torch.manual_seed(rand_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(rand_seed)
    return rand_seed


def run(args):
    mode = args.parse_string("mode", "train")
    num_envs = args.parse_int("num_envs", 1)
    device = args.parse_string("device", "cuda:0")
    if "LOCAL_RANK" in os.environ and device.startswith("cuda"):
        device = "cuda:{}".format(int(os.environ["LOCAL_RANK"]))
        torch.cuda.set_device(device)
    dist_util.init_from_env(device)
    visualize = args.parse_bool("visualize", False)
    logger_type = args.parse_string("logger", "tb")
    log_file = args.parse_string("log_file", "output/log.txt")
    out_model_file = args.parse_string("out_model_file", "output/model.pt")
    int_output_dir = args.parse_string("int_output_dir", "")
    model_file = args.parse_string("model_file", "")
    seed = set_rand_seed(args)
    np.set_printoptions(edgeitems=30, linewidth=4000, precision=2, threshold=10000)
    for d in (os.path.dirname(out_model_file), int_output_dir):
        if d != "":
            os.makedirs(d, exist_ok=True)

    ws, rk = dist_util.world_size(), dist_util.rank()
    env = env_builder.build_env(path_loader.resolve_path(args.parse_string("env_config")), num_envs, device, visualize,
                                env_id_base=rk * num_envs, total_envs=ws * num_envs, seed=seed)
    agent = agent_builder.build_agent(path_loader.resolve_path(args.parse_string("agent_config")), env, device)
    if model_file != "":
        agent.load(path_loader.resolve_path(model_file))

    if mode == "train":
        max_samples = args.parse_int("max_samples", np.iinfo(np.int64).max)
        agent.train_model(max_samples=max_samples, out_model_file=out_model_file, int_output_dir=int_output_dir,
                          logger_type=logger_type, log_file=log_file)
    elif mode == "test":
        result = agent.test_model(num_episodes=args.parse_int("test_episodes", 32))
        Logger.print("Mean Return: {}".format(result["mean_return"]))
        Logger.print("Mean Episode Length: {}".format(result["mean_ep_len"]))
        Logger.print("Episodes: {}".format(result["num_eps"]))
        for key in result:  # run_tracker.py:54-56: the tracking-error means when the env reports them
            if "test_mean" in key:
                Logger.print(key + ": {}".format(result[key]))
    elif mode == "record":  # PARC stage 4 (parc_4_phys_record.py -> run_tracker.run mode record -> record_dm_motions)
        from parc_amd.learning.dm_motion_recorder import record_dm_motions
        record_dm_motions(agent)
    else:
        raise AssertionError("Unsupported mode: {}".format(mode))
    if hasattr(env, "check_health"):  # hand-off timeouts of the dynamics kernel: raises -> non-zero exit code
        env.check_health()


if __name__ == "__main__":
    run(load_args(sys.argv))
