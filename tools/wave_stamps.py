#!/usr/bin/env python3
"""Developer tool: per-wave timeline of k_dynamics_wave from a -DPARC_STAMPS build (PARC_ENV_LIB=variants/libparc_env_stamps.so).
Prints, per wave role, the mean cycles per control step spent in each work segment and in the barrier wait after it."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd.envs.hip_parkour_env import HipParkourEnv
from parc_amd.util import path_loader

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = path_loader.load_config("data/configs/tracker_config/dm_env_default.yaml")
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1, mirror_ref_state=False, enable_dynamics=True)
env.reset()
lo, hi = env._action_bound_low, env._action_bound_high
torch.manual_seed(0)
ACT = (0.5 * (hi + lo) + 0.05 * 0.5 * (hi - lo) * torch.randn(n, env._char_dof_pos.shape[1], device="cuda:0")).contiguous()
for _ in range(20):
    env.step(ACT); env.reset_done()
torch.cuda.synchronize()
a = (C.c_double * 64)()
env._lib.parc_env_debug_wave_stamps.argtypes = [C.POINTER(C.c_double)]
env._lib.parc_env_debug_wave_stamps(a)  # clear
_c = (C.c_double * 128)()
env._lib.parc_env_debug_wave_counts.argtypes = [C.POINTER(C.c_double)]
env._lib.parc_env_debug_wave_counts(_c)  # clear
iters = 20
for _ in range(iters):
    env.step(ACT); env.reset_done()
torch.cuda.synchronize()
env._lib.parc_env_debug_wave_stamps(a)
nblk = (n + 63) // 64
names = ["prologue", "P1 trunk FK", "wait1", "P2A limbs", "(plane eval)", "P2B", "(segments)", "P3 root", "wait4", "P4 limbs out", "epilogue",
         "(limb FK)", "(own inertia)", "(narrow points)", "(joint inward)", "(misc)"]
# the bracketed slots are carved out of P2A / P2B / P3 (finer stamps inside the bodies); the phase slots then hold the remainder
out = {}
for w in range(4):
    row = {names[i]: round(a[w * 16 + i] / (nblk * iters)) for i in range(16)}
    row["total"] = sum(row.values())
    out[f"wave{w}"] = row
print(json.dumps({"envs": n, "cycles_per_control_step": out}, indent=1))
c = (C.c_double * 128)()
env._lib.parc_env_debug_wave_counts.argtypes = [C.POINTER(C.c_double)]
env._lib.parc_env_debug_wave_counts(c)
bn = env._kin_char_model.get_body_names()
# -DPARC_COUNTS builds: how much of the contact narrow phase a wave executes vs how many lanes need it
print("%-18s %10s %10s %12s %14s %8s" % ("body", "near lanes", "near waves", "lane use", "executed/cands", "slow"), file=sys.stderr)
for b in range(len(bn)):
    r = [c[b * 8 + i] for i in range(8)]
    if r[0] == 0: continue
    # lane use = (lane, candidate) pairs that needed the narrow phase / (64 lanes x executed candidates)
    print("%-18s %10.4f %10.4f %12.4f %8.2f / %-4.0f %8.5f" % (bn[b], r[1] / r[0], r[2] / r[7], r[3] / max(64.0 * r[6], 1.0), r[6] / r[7], r[4] / r[7], r[5] / r[0]), file=sys.stderr)
