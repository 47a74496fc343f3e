"""Calibration script of tests/test_hip_parity.py::test_dynamics_kernel_matches_cpu_build (lives under tests/ because it uses the CPU oracle,
which only the tests, smoke() and bench.py's cpu_baseline leg may touch): k_dynamics_wave vs the host build of the same equations (oracle/dyn_oracle.cpp), per-env error statistics and what the
outlier envs have in common (calibrates tests/test_hip_parity.py::test_dynamics_kernel_matches_cpu_build)."""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from gpu_helpers import default_config, write_motion_yaml, to_np
from helpers import CLIPS4
from oracle.binding_dyn import DynOracle
from parc_amd.envs.hip_parkour_env import HipParkourEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tmp = pathlib.Path(tempfile.mkdtemp())
cfg = default_config()
cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp, CLIPS4, [1, 1, 1, 1])
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=5, enable_dynamics=True)
env.reset()
d = DynOracle(env._scene.cfg)
sc = env._scene
hf, mp, dxdy = sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy
gen = torch.Generator(device="cuda:0"); gen.manual_seed(12)
TOL = [("root_pos", "_char_root_pos", 2e-4), ("root_rot", "_char_root_rot", 2e-4), ("root_vel", "_char_root_vel", 5e-3),
       ("root_ang_vel", "_char_root_ang_vel", 2e-2), ("dof_pos", "_char_dof_pos", 1e-3), ("dof_vel", "_char_dof_vel", 5e-2)]
for it in range(4):
    act = (env._char_dof_pos + 0.1 * torch.randn(env._char_dof_pos.shape, device="cuda:0", generator=gen)).contiguous()
    st = dict(root_pos=to_np(env._char_root_pos).copy(), root_rot=to_np(env._char_root_rot).copy(), root_vel=to_np(env._char_root_vel).copy(),
              root_ang_vel=to_np(env._char_root_ang_vel).copy(), dof_pos=to_np(env._char_dof_pos).copy(), dof_vel=to_np(env._char_dof_vel).copy(),
              contact_force=np.zeros((n, 15, 3), np.float32))
    env.step(act)
    d.step(hf, mp, dxdy, st, to_np(act), sc.env_offsets)
    fg, fc = to_np(env._char_contact_forces), st["contact_force"]
    cg, cc = np.linalg.norm(fg, axis=-1) > 1e-5, np.linalg.norm(fc, axis=-1) > 1e-5
    set_differs = (cg != cc).any(1)
    # relative force difference per body where both are in contact
    fdiff = np.abs(fg - fc).reshape(n, -1).max(1)
    worst = np.zeros(n)
    for k_o, k_e, tol in TOL:
        err = np.abs(to_np(getattr(env, k_e)) - st[k_o]).reshape(n, -1).max(1)
        worst = np.maximum(worst, err / tol)
        out = err > 20 * tol
        print(it, k_o, "q99 %.2e q999 %.2e max %.2e | outliers(>20tol) %.4f of which contact-set differs %.3f | outlier max %.3e | non-outlier q999/tol %.2f" % (
            np.quantile(err, 0.99), np.quantile(err, 0.999), err.max(), out.mean(), set_differs[out].mean() if out.any() else -1, err[out].max() if out.any() else 0,
            np.quantile(err[~out], 0.999) / tol))
    out = worst > 20
    print(it, "ANY-key outliers %.4f; contact-set differs among them %.3f; among all envs %.4f; force diff of outliers w/o set change: %s" % (
        out.mean(), set_differs[out].mean() if out.any() else -1, set_differs.mean(), np.round(fdiff[out & ~set_differs][:8], 2)))
