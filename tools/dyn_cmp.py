"""Dev tool: one control step of two dynamics kernels (PARC_DYN_KERNEL values given on the command line) from the same state."""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from gpu_helpers import default_config, write_motion_yaml, to_np
from parc_amd.envs.hip_parkour_env import HipParkourEnv

ka, kb = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
p = torch.cuda.get_device_properties(0)
print("shared per block", p.shared_memory_per_block, getattr(p, "shared_memory_per_block_optin", None))
envs = []
tmp = pathlib.Path(tempfile.mkdtemp())
for k in (ka, kb):
    os.environ["PARC_DYN_KERNEL"] = k
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp, ["civilization", "sfu"], [1.0, 1.0])
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=9, enable_dynamics=True, mirror_ref_state=True)
    env.reset()
    envs.append(env)
names = ["_char_root_pos", "_char_root_rot", "_char_root_vel", "_char_root_ang_vel", "_char_dof_pos", "_char_dof_vel", "_char_contact_forces"]
for it in range(3):
    act = (envs[0]._char_dof_pos + 0.1 * torch.randn_like(envs[0]._char_dof_pos)).contiguous()
    for nm in names:
        getattr(envs[1], nm).copy_(getattr(envs[0], nm))
    if it == 0 and os.environ.get("DYN_CMP_DUMP"):
        np.savez(os.environ["DYN_CMP_DUMP"], act=to_np(act), env_offsets=envs[0]._scene.env_offsets, **{nm: to_np(getattr(envs[0], nm)).copy() for nm in names})
    for env in envs:
        env.step(act)
    if it == 0 and os.environ.get("DYN_CMP_DUMP"):
        np.savez(os.environ["DYN_CMP_DUMP"].replace(".npz", "_after.npz"), **{ka + nm: to_np(getattr(envs[0], nm)).copy() for nm in names}, **{kb + nm: to_np(getattr(envs[1], nm)).copy() for nm in names})
    for nm in names:
        a, b = to_np(getattr(envs[0], nm)).reshape(n, -1), to_np(getattr(envs[1], nm)).reshape(n, -1)
        err = np.abs(a - b)
        bad = np.where(~np.isfinite(err).all(1) | (err.max(1) > 1e-3))[0]
        print(it, nm, "max", np.nanmax(err), "nan rows", int((~np.isfinite(a)).any(1).sum()), int((~np.isfinite(b)).any(1).sum()), "bad rows", len(bad), bad[:12])
        if nm == "_char_contact_forces":
            worst = int(np.argmax(err.max(1)))
            names = envs[0]._scene.char_model.get_body_names()
            fa, fb = a[worst].reshape(15, 3), b[worst].reshape(15, 3)
            print("   worst env", worst, "force err", err[worst].max())
            for bd in range(15):
                if np.abs(fa[bd]).max() > 0 or np.abs(fb[bd]).max() > 0:
                    print("     %-16s %s: %s   %s: %s" % (names[bd], ka, np.round(fa[bd], 2), kb, np.round(fb[bd], 2)))
