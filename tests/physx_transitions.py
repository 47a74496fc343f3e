"""TEST INFRASTRUCTURE (tests/test_dynamics_cpu.py::test_physx_recorded_transitions_*, tools/physx_onestep.py): the PhysX-anchored check of the
re-authored dynamics that needs no policy.

`dec2024_teaser_717_1_opt_dm.pkl` holds, per 30 Hz control step, the pose PhysX produced, the bodies that carried a contact force and -- in
`misc_data["obs"]`, the observation stream the policy saw (ig_parkour_env.py:698-736) -- the root and dof VELOCITIES PhysX reported.  The PD
targets are not recorded.  For every recorded transition s_t -> s_t+1:
  * start the host build of the simulator AT the recorded state s_t (pose from the frames, velocities from the obs row; the four arm
    joints' obs entries use an older XML convention, their velocities are central differences of the frames);
  * the 28 PD targets are the only unknown of the control step: take the targets that make THIS simulator reproduce the recorded next joint
    angles (fixed-point iteration a <- a + (q_rec - q_sim), a few control steps of the host build per transition) -- the joints are then
    on the recorded path by construction, and what is left is what no target can buy: the motion of the unactuated floating base
    (6 dofs: it moves as gravity, the contact forces and the joint reactions dictate) and the set of touching bodies;
  * `one_step`: errors of the root (position, height, rotation, velocity) against the recorded s_t+1, next to null models (root continues
    with its recorded velocity; free fall; no motion), and the agreement of the contact flags;
  * `closed_loop(h)`: the SIMULATED state is fed back for h control steps (targets refitted each step from the simulated state: a deadbeat
    joint-space tracking controller) and the root drift is reported -- the closed-loop check the open-loop replay could not be.
"""
import os

import numpy as np

ARM_JOINTS = (2, 3, 5, 6)   # their recorded obs rows use an older XML convention (tests/test_oracle_golden.py::test_recorded_isaacgym_obs)


def load(oracle, orc_char, cm_golden):
    from conftest import DATA, golden
    from parc_amd import ms_file
    f = ms_file.load_ms_file(os.path.join(DATA, "motion_terrains", "dec2024_teaser_717_1_opt_dm.pkl"))
    md, td = f.motion_data, f.terrain_data
    obs = golden("recorded_obs_dec2024_teaser_717_1_opt_dm")["obs"]
    n = md.root_pos.shape[0]
    dt = 1.0 / float(md.fps)
    rp, rr = md.root_pos.astype(np.float32), md.root_rot.astype(np.float32)
    dof = oracle.rot_to_dof(orc_char, np.ascontiguousarray(md.joint_rot, np.float32))
    # velocities: obs[:, 6:9] / [9:12] are the root velocities in the heading frame (compute_char_obs, ig_char_env.py:593-597), [96:124] the dof velocities
    heading = oracle.calc_heading(rr)
    c, s = np.cos(heading), np.sin(heading)
    def to_world(v):
        return np.stack([c * v[:, 0] - s * v[:, 1], s * v[:, 0] + c * v[:, 1], v[:, 2]], axis=-1).astype(np.float32)
    rv, rw = to_world(obs[:, 6:9]), to_world(obs[:, 9:12])
    dv = obs[:, 96:124].astype(np.float32).copy()
    jt, di = cm_golden["joint_type"], cm_golden["dof_idx"]
    arm_dofs = []
    for j in ARM_JOINTS:
        b = j + 1
        arm_dofs += list(range(int(di[b]), int(di[b]) + {1: 1, 2: 3}[int(jt[b])]))
    fd = np.zeros_like(dof); fd[1:-1] = (dof[2:] - dof[:-2]) / (2 * dt); fd[0] = (dof[1] - dof[0]) / dt; fd[-1] = (dof[-1] - dof[-2]) / dt
    dv[:, arm_dofs] = fd[:, arm_dofs]      # arm joints: central differences of the frames instead of the old-convention obs entries
    return dict(n=n, dt=dt, root_pos=rp, root_rot=rr, root_vel=rv, root_ang_vel=rw, dof=dof, dof_vel=dv, contacts=md.body_contacts > 0.5,
                hf=td.hf, min_point=td.min_point, dx=(td.dx, td.dx), fd_check=(fd, arm_dofs))


def sim_step(d, R, st, act):
    n = st["root_pos"].shape[0]
    d.step(R["hf"], R["min_point"], R["dx"], st, act, np.zeros((n, 3), np.float32))


def fit_and_step(d, R, st0, q_next, iters=8):
    """targets that put the simulated next joint angles on q_next; returns (state after the step, targets, residual)."""
    act = q_next.copy()
    st = None
    for _ in range(iters):
        st = {k: v.copy() for k, v in st0.items()}
        sim_step(d, R, st, act)
        act = (act + (q_next - st["dof_pos"])).astype(np.float32)
    st = {k: v.copy() for k, v in st0.items()}
    sim_step(d, R, st, act)
    return st, act, np.abs(st["dof_pos"] - q_next).max(axis=1)


def quat_angle(oracle, a, b):
    return np.abs(oracle.quat_diff_angle(a, b))


def one_step(d, oracle, R, gravity_only=False):
    n = R["n"]
    idx = np.arange(n - 1)
    st0 = dict(root_pos=R["root_pos"][idx].copy(), root_rot=R["root_rot"][idx].copy(), root_vel=R["root_vel"][idx].copy(), root_ang_vel=R["root_ang_vel"][idx].copy(),
               dof_pos=R["dof"][idx].copy(), dof_vel=R["dof_vel"][idx].copy(), contact_force=np.zeros((n - 1, 15, 3), np.float32))
    st, act, res = fit_and_step(d, R, st0, R["dof"][idx + 1])
    nxt = idx + 1
    e_pos = np.linalg.norm(st["root_pos"] - R["root_pos"][nxt], axis=1)
    e_z = np.abs(st["root_pos"][:, 2] - R["root_pos"][nxt, 2])
    e_rot = quat_angle(oracle, st["root_rot"], R["root_rot"][nxt])
    e_vel = np.linalg.norm(st["root_vel"] - R["root_vel"][nxt], axis=1)
    # null models for the root position: constant velocity, and free fall with the recorded velocity
    cv = R["root_pos"][idx] + R["dt"] * R["root_vel"][idx]
    ff = cv + np.array([0, 0, -0.5 * 9.81 * R["dt"] ** 2], np.float32)
    e_cv = np.linalg.norm(cv - R["root_pos"][nxt], axis=1)
    e_ff = np.linalg.norm(ff - R["root_pos"][nxt], axis=1)
    e_hold = np.linalg.norm(R["root_pos"][idx] - R["root_pos"][nxt], axis=1)
    sim_c = np.linalg.norm(st["contact_force"], axis=-1) > 1e-5
    ref_c = R["contacts"][nxt] | R["contacts"][idx]     # a body that touched at either end of the step
    agree = (sim_c == R["contacts"][nxt])
    feet = [11, 14]
    fz = st["contact_force"][:, :, 2].sum(1)
    return dict(fit_residual=res, e_pos=e_pos, e_z=e_z, e_rot=e_rot, e_vel=e_vel, e_cv=e_cv, e_ff=e_ff, e_hold=e_hold, agree_all=agree.mean(), agree_feet=agree[:, feet].mean(),
                foot_rate_sim=sim_c[:, feet].mean(), foot_rate_ref=R["contacts"][nxt][:, feet].mean(), false_neg=(~sim_c & R["contacts"][nxt] & R["contacts"][idx]).mean(),
                fz=fz, act=act, ref_any=ref_c)


def closed_loop(d, oracle, R, horizon=10, stride=3):
    """from every `stride`-th recorded state: `horizon` control steps on the SIMULATED state, targets refitted each step."""
    n = R["n"]
    starts = np.arange(0, n - 1 - horizon, stride)
    st = dict(root_pos=R["root_pos"][starts].copy(), root_rot=R["root_rot"][starts].copy(), root_vel=R["root_vel"][starts].copy(), root_ang_vel=R["root_ang_vel"][starts].copy(),
              dof_pos=R["dof"][starts].copy(), dof_vel=R["dof_vel"][starts].copy(), contact_force=np.zeros((len(starts), 15, 3), np.float32))
    out = []
    for h in range(1, horizon + 1):
        st, act, res = fit_and_step(d, R, st, R["dof"][starts + h], iters=6)
        e_pos = np.linalg.norm(st["root_pos"] - R["root_pos"][starts + h], axis=1)
        e_z = np.abs(st["root_pos"][:, 2] - R["root_pos"][starts + h, 2])
        e_rot = quat_angle(oracle, st["root_rot"], R["root_rot"][starts + h])
        sim_c = np.linalg.norm(st["contact_force"], axis=-1) > 1e-5
        agree = (sim_c == R["contacts"][starts + h])
        out.append(dict(h=h, e_pos_med=float(np.median(e_pos)), e_pos_q90=float(np.quantile(e_pos, 0.9)), e_z_med=float(np.median(e_z)), e_rot_med=float(np.median(e_rot)),
                        e_rot_q90=float(np.quantile(e_rot, 0.9)), agree_all=float(agree.mean()), agree_feet=float(agree[:, [11, 14]].mean()),
                        foot_rate_sim=float(sim_c[:, [11, 14]].mean()), foot_rate_ref=float(R["contacts"][starts + h][:, [11, 14]].mean()), fit_res=float(res.max())))
    return out


