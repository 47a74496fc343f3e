"""Dev tool: run the bench scenario / settle run of tests/test_dynamics_gpu.py step by step; at the first non-finite state dump the
state BEFORE that step of the offending envs (npz) for an offline reproduction with the host build."""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import test_dynamics_gpu as T
from gpu_helpers import to_np
from parc_amd.envs.hip_parkour_env import HipParkourEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
motions = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tmp = pathlib.Path(tempfile.mkdtemp())
env = HipParkourEnv(T._cfg3(tmp, motions), n, "cuda:0", False, seed=21, enable_dynamics=True, mirror_ref_state=False)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(4)
env.reset()
names = ["_char_root_pos", "_char_root_rot", "_char_root_vel", "_char_root_ang_vel", "_char_dof_pos", "_char_dof_vel"]


def snap():
    return {nm: getattr(env, nm).clone() for nm in names}


VMAX = float(os.environ.get("HUNT_VMAX", "25"))


def check(tag, it, before, act):
    bad = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    for nm in names:
        bad |= ~torch.isfinite(getattr(env, nm)).reshape(n, -1).all(1)
    bad |= env._char_root_vel.norm(dim=-1) > VMAX      # a launch: nothing in these scenarios moves that fast (a 13 m fall: 16 m/s)
    if bad.any():
        ids = torch.nonzero(bad).flatten()[:16].cpu().numpy()
        print(tag, "step", it, "non-finite envs:", int(bad.sum()), ids, "timeouts", env.dynamics_timeouts(), flush=True)
        np.savez("gpurun_out/nan_state.npz", ids=ids, act=to_np(act)[ids], env_offsets=env._scene.env_offsets[ids],
                 **{nm: to_np(before[nm])[ids] for nm in names})
        for nm in names:
            print("  before", nm, to_np(before[nm])[ids[0]])
        for nm in names:
            print("  after ", nm, to_np(getattr(env, nm))[ids[0]])
        print("  contact forces after", to_np(env._char_contact_forces)[ids[0]])
        sys.exit(0)


for it in range(30):
    before = snap()
    act = T._bench_actions(env, gen)
    env.step(act)
    check("bench", it, before, act)
    env.reset_done()
env.reset()
env._char_root_vel.zero_(); env._char_root_ang_vel.zero_(); env._char_dof_vel.zero_()
hold = env._char_dof_pos.clone()
for it in range(240):
    before = snap()
    env.step(hold)
    check("settle", it, before, hold)
print("no non-finite state; max |root_vel|", float(env._char_root_vel.abs().max()), "timeouts", env.dynamics_timeouts())
