"""Dev tool: open-loop replay of the PhysX-recorded clip (tests/test_dynamics_gpu.py::physx_replay_metrics) with a per-start timeline:
when and why each start terminates, what its feet do against the recorded contact flags, and the terrain under them.

    python tools/replay_diag.py [n_starts] [verbose env ids ...]
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import test_dynamics_gpu as T
from gpu_helpers import default_config, to_np
from conftest import DATA
from parc_amd import lib as L
from parc_amd.envs.hip_parkour_env import HipParkourEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
verbose = [int(a) for a in sys.argv[2:]]
clip = "dec2024_teaser_717_1_opt_dm"
m = T.physx_replay_metrics(n)
print({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in m.items()})

cfg = default_config()
cfg["env"]["dm"]["motion_file"] = os.path.join(DATA, "motion_terrains", clip + ".pkl")
cfg["env"]["dm"]["terrain_build_mode"] = "file"
cfg["env"]["rand_reset"] = False
cfg["env"]["rand_root_pos_offset_scale"] = 0.0
env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1, enable_dynamics=True, mirror_ref_state=True)
env.set_reset_motion_start_time_fraction(torch.linspace(0.0, 0.8, n, device="cuda:0"))
env.reset()
sc = env._scene; ter = sc.grid.terrain; hf = ter.hf
print("terrain", hf.shape, "heights", np.unique(np.round(hf, 2))[:20], "min_point", ter.min_point, "dx", ter.dxdy)
dt = 1.0 / 30.0
length = float(env._motion_lengths[0]); nsteps = int(round(length / dt))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ids = torch.zeros(n, dtype=torch.int32, device="cuda:0")
z = lambda *s: torch.zeros(*s, device="cuda:0")
o = dict(root_pos=z(n, 3), root_rot=z(n, 4), root_vel=z(n, 3), root_ang_vel=z(n, 3), joint_rot=z(n, 14, 4), dof_vel=z(n, 28), contacts=z(n, 15))
target = z(n, 28)
names = sc.char_model.get_body_names()
term = np.array(sc.cfg.pose_termination_dist[:14])


def ground(p):  # env-local [n, 3] -> terrain height under it
    g = p[:, :2] + sc.env_offsets[:, :2]
    ix = np.clip(np.rint((g[:, 0] - ter.min_point[0]) / ter.dxdy[0]).astype(int), 0, hf.shape[0] - 1)
    iy = np.clip(np.rint((g[:, 1] - ter.min_point[1]) / ter.dxdy[1]).astype(int), 0, hf.shape[1] - 1)
    return hf[ix, iy]


failed = np.zeros(n, bool)
for k in range(nsteps):
    t_next = (env._timestep_buf.float() + 1.0) * dt + env._motion_time_offsets
    L.check(env._lib.parc_calc_motion_frame(env._handle, ids.data_ptr(), t_next.contiguous().data_ptr(), n,
                                            *[o[q].data_ptr() for q in ["root_pos", "root_rot", "root_vel", "root_ang_vel", "joint_rot", "dof_vel", "contacts"]], st))
    L.check(env._lib.parc_rot_to_dof(env._handle, o["joint_rot"].data_ptr(), target.data_ptr(), n, st))
    _, rew, done, _ = env.step(target)
    d = to_np(done)
    bp, rbp = to_np(env._char_rigid_body_pos), to_np(env._ref_body_pos)
    rel = np.linalg.norm((bp - bp[:, :1]) - (rbp - rbp[:, :1]), axis=-1)[:, 1:]          # body-vs-root offsets, what compute_done tests
    rootd = np.linalg.norm(to_np(env._char_root_pos) - to_np(env._ref_root_pos), axis=1)
    simc = to_np(env._char_contact_forces.norm(dim=-1) > 1e-5); refc = to_np(env._ref_contacts) > 0.5
    fz = to_np(env._char_contact_forces)[:, :, 2]
    mt = to_np(env._timestep_buf.float() * dt + env._motion_time_offsets)
    for i in range(n):
        if failed[i] or mt[i] >= length - 1e-4:
            continue
        if i in verbose:
            print("  env %2d k %3d t %.2f rootd %.3f zerr %+.3f worst body %-16s %.3f/%.2f | feet sim %d%d ref %d%d fz %6.0f %6.0f | foot clearance %.3f %.3f ref %.3f %.3f | rew %.3f" % (
                i, k, mt[i], rootd[i], to_np(env._char_root_pos)[i, 2] - to_np(env._ref_root_pos)[i, 2], names[1 + int(np.argmax(rel[i] / term))], rel[i].max(), term[int(np.argmax(rel[i] / term))],
                simc[i, 11], simc[i, 14], refc[i, 11], refc[i, 14], fz[i, 11], fz[i, 14],
                bp[i, 11, 2] + sc.env_offsets[i, 2] - ground(bp[:, 11])[i], bp[i, 14, 2] + sc.env_offsets[i, 2] - ground(bp[:, 14])[i],
                rbp[i, 11, 2] + sc.env_offsets[i, 2] - ground(rbp[:, 11])[i], rbp[i, 14, 2] + sc.env_offsets[i, 2] - ground(rbp[:, 14])[i], float(rew[i])))
        if d[i] == 1:
            failed[i] = True
            why = []
            if (rel[i] > term).any():
                why.append("pose: " + ", ".join("%s %.2f>%.2f" % (names[1 + j], rel[i, j], term[j]) for j in np.nonzero(rel[i] > term)[0]))
            if rootd[i] > 0.6:
                why.append("root pos %.2f" % rootd[i])
            print("env %2d start t0 %.2f fails at step %3d (clip t %.2f): %s | contacts sim %s ref %s" % (
                i, float(env._motion_time_offsets[i]), k, mt[i], "; ".join(why) or "root rot / other",
                "".join(str(int(c)) for c in simc[i]), "".join(str(int(c)) for c in refc[i])))
