"""Rank body for the launcher test (tests/test_bench_launcher_cpu.py): gloo rendezvous from the env vars bench.launch_ranks sets,
one all-reduce, rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
dist.barrier()
if dist.get_rank() == 0:
    print(json.dumps({"world": dist.get_world_size(), "sum": float(t.item()), "argv": sys.argv[1:],
                      "local_rank": os.environ["LOCAL_RANK"], "addr": os.environ["MASTER_ADDR"]}))
dist.destroy_process_group()
