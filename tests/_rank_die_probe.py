"""Rank body for the launcher's failure-path test: every rank joins the gloo group (so the rendezvous HAS happened), then rank 1
exits with code 3 while the others enter a collective that can no longer complete."""
import os
import sys

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
dist.barrier()
if dist.get_rank() == 1:
    os._exit(3)
t = torch.ones(1)
dist.all_reduce(t)       # hangs: rank 1 is gone
dist.barrier()
print("unreachable")
sys.exit(0)
