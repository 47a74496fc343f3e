"""Dev tool: settle-to-rest on the cfg-3 scene (hold the reset pose for 4 s) and a CLASSIFICATION of the envs whose contact forces do
not carry the weight within 25 % at the end: what are they doing?  Calibrates tests/test_dynamics_gpu.py.

    python tools/dyn_settle_diag.py [n_envs] [out.json]
"""
import json, os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import test_dynamics_gpu as T
from gpu_helpers import to_np
from parc_amd.envs.hip_parkour_env import HipParkourEnv
tmp = pathlib.Path(tempfile.mkdtemp())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = HipParkourEnv(T._cfg3(tmp, 1024), n, "cuda:0", False, seed=21, enable_dynamics=True, mirror_ref_state=False)
env.reset()
env._char_root_vel.zero_(); env._char_root_ang_vel.zero_(); env._char_dof_vel.zero_()
hold = env._char_dof_pos.clone()
mg = 9.81 * 50.05
mid = to_np(env._motion_ids) % 5
z0 = to_np(env._char_root_pos)[:, 2].copy()
sc = env._scene
ter = sc.grid.terrain
hf = ter.hf


def ground_under(pos_local):
    """terrain height under env-local points [n, k, 3] (nearest cell, like the reference)."""
    g = pos_local[..., :2] + sc.env_offsets[:, None, :2]
    ix = np.clip(np.rint((g[..., 0] - ter.min_point[0]) / ter.dxdy[0]).astype(int), 0, hf.shape[0] - 1)
    iy = np.clip(np.rint((g[..., 1] - ter.min_point[1]) / ter.dxdy[1]).astype(int), 0, hf.shape[1] - 1)
    return hf[ix, iy]


last = []
hist_f = []
T_END = int(os.environ.get("SETTLE_STEPS", "120"))
for it in range(T_END):
    env.step(hold)
    if it >= T_END - 10:
        last.append(to_np(env._char_contact_forces)[:, :, 2].sum(1) / mg)
        hist_f.append(to_np(env._char_contact_forces).copy())
    if it % 15 == 14:
        f = to_np(env._char_contact_forces)
        fz = f[:, :, 2].sum(1) / mg
        sp = to_np(env._char_root_vel.norm(dim=-1))
        fmax = np.abs(f).reshape(n, -1).max(1) / mg
        print("t=%.1fs fz med %.3f |fz-1|<.25: %.3f fz==0: %.3f fz>3: %.4f fmax>20mg: %.4f speed med %.3f q90 %.3f q99 %.2f  dz med %.2f q01 %.2f" % (
            (it + 1) / 30, np.median(fz), np.mean(np.abs(fz - 1) < 0.25), np.mean(fz == 0), np.mean(fz > 3), np.mean(fmax > 20), np.median(sp), np.quantile(sp, 0.9),
            np.quantile(sp, 0.99), np.median(to_np(env._char_root_pos)[:, 2] - z0), np.quantile(to_np(env._char_root_pos)[:, 2] - z0, 0.01)), flush=True)
last = np.stack(last)                      # [10, n] : sum Fz / mg over the last 10 control steps
fz = last[-1]
f = to_np(env._char_contact_forces)
ncont = (np.linalg.norm(f, axis=-1) > 1e-5).sum(1)
rv = to_np(env._char_root_vel); sp = np.linalg.norm(rv, axis=1)
dofv = np.abs(to_np(env._char_dof_vel)).max(1)
bp = to_np(env._char_rigid_body_pos)
clear = (bp[..., 2] + sc.env_offsets[:, None, 2] - ground_under(bp)).min(1)   # lowest body ORIGIN above the ground under it
root_cell_dist = np.abs(bp[..., :2] - bp[:, :1, :2]).max((1, 2)) / float(ter.dxdy[0])   # farthest body origin from the root, in cells (patch: +-4.5)
out = np.abs(fz - 1.0) >= 0.25
mean10, std10 = last.mean(0), last.std(0)
cls = np.full(n, "", dtype=object)
cls[out & (mean10 == 0) & (rv[:, 2] < -1.0)] = "falling (no contact, vz < -1 m/s)"
cls[out & (mean10 == 0) & (rv[:, 2] >= -1.0)] = "no contact, not falling"
cls[out & (cls == "") & (np.abs(mean10 - 1.0) < 0.1)] = "supported on average (|mean10 - 1| < 0.1): force chatter"
cls[out & (cls == "") & (sp > 0.5)] = "in contact, moving (> 0.5 m/s): tipping / sliding / bouncing"
cls[out & (cls == "") & (mean10 > 1.1)] = "over-supported at low speed (mean10 > 1.1)"
cls[out & (cls == "") & (mean10 < 0.9)] = "under-supported at low speed (mean10 < 0.9)"
cls[out & (cls == "")] = "other"
rows = []
print("\nenvs outside +-25 %% at 4 s: %d of %d (%.3f)" % (out.sum(), n, out.mean()))
print("%-62s %6s %6s | %6s %6s %6s %6s %6s %6s %6s | clips" % ("class", "count", "share", "fz", "std10", "speed", "dofv", "ncont", "clear", "cells"))
for c in sorted(set(cls[out]), key=lambda c: -(cls == c).sum()):
    m = cls == c
    row = dict(cls=c, count=int(m.sum()), share_of_all=float(m.mean()), fz_med=float(np.median(fz[m])), std10_med=float(np.median(std10[m])),
               speed_med=float(np.median(sp[m])), dof_vel_med=float(np.median(dofv[m])), ncontact_med=float(np.median(ncont[m])),
               lowest_origin_clearance_med=float(np.median(clear[m])), farthest_body_cells_max=float(root_cell_dist[m].max()),
               per_clip=[int((m & (mid == k)).sum()) for k in range(5)])
    rows.append(row)
    print("%-62s %6d %6.4f | %6.2f %6.2f %6.2f %6.1f %6.1f %6.2f %6.2f | %s" % (c, row["count"], row["share_of_all"], row["fz_med"], row["std10_med"], row["speed_med"],
          row["dof_vel_med"], row["ncontact_med"], row["lowest_origin_clearance_med"], row["farthest_body_cells_max"], row["per_clip"]))
inside = ~out
print("inside: fz med %.3f std10 med %.3f speed med %.3f ncont med %.1f" % (np.median(fz[inside]), np.median(std10[inside]), np.median(sp[inside]), np.median(ncont[inside])))
print("clips:", [c.name for c in sc.clips[:5]], "envs per clip", [int((mid == k).sum()) for k in range(5)])
names = sc.char_model.get_body_names()
chat = np.nonzero(out & (std10 > 0.3) & (sp < 1.0))[0]
print("\nforce chatter detail (last 6 control steps, bodies in contact; forces in units of mg): %d envs" % len(chat))
hist_f = np.stack(hist_f) / mg            # [10, n, 15, 3]
touched = (np.linalg.norm(hist_f[:, chat], axis=-1) > 1e-5 / mg).any(0).sum(0)
print("bodies touching in the chatter envs:", {names[b]: int(touched[b]) for b in range(15) if touched[b]})
for e in chat[:8]:
    print(" env", int(e), "clip", int(mid[e]), "speed %.2f" % sp[e], "root z-z0 %.2f" % (to_np(env._char_root_pos)[e, 2] - z0[e]))
    for k in range(4, 10):
        fb = hist_f[k, e]
        act = np.nonzero(np.linalg.norm(fb, axis=-1) > 1e-6)[0]
        print("   step %d: " % k + "  ".join("%s(%.2f %.2f %.2f)" % (names[b][:10], fb[b, 0], fb[b, 1], fb[b, 2]) for b in act))
print("timeouts", env.dynamics_timeouts())
if len(sys.argv) > 2:
    json.dump(dict(n=n, outside=int(out.sum()), rows=rows, inside_fz_med=float(np.median(fz[inside])), inside_std10_med=float(np.median(std10[inside]))),
              open(sys.argv[2], "w"), indent=1)
