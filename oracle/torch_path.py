"""PyTorch-CPU restatement of the reference's motion_lib / kin_char_model path.  TEST INFRASTRUCTURE ONLY (imported by
tests/ and bench.py's cpu_baseline leg; nothing under parc_amd/ touches it).

BASELINE.json's north star asks for the GPU numbers "next to the reference CPU-PyTorch motion_lib/kin_char_model path timed
on the same box's host cores".  The reference itself cannot travel to the GPU box, so this file issues the SAME batched
tensor-op sequence on CPU tensors — one ATen op per arithmetic step, as the reference's TorchScript functions do:

  * quaternion ops                torch_util.py:31-67, 70-91, 337-342, 426-462, 475-499
  * MotionLib.calc_motion_frame   motion_lib.py:94-128 (+ _calc_frame_blend :425-438, calc_phase :520, loop offset :440-460)
  * KinCharModel.dof_to_rot       kin_char_model.py:586-599  (python loop over the 14 joints)
  * KinCharModel.rot_to_dof       kin_char_model.py:601-615
  * KinCharModel.forward_kinematics kin_char_model.py:617-649 (python loop over the 15 bodies)

and `step_path` strings them together the way one env step uses them: the reference pose at t, the six
look-ahead targets, FK of the character, of the reference and of the targets, dof <-> rot of the character / reference.
Pinned by tests/test_oracle_golden.py::test_torch_path_vs_golden against the golden vectors the real reference produced
(tests/golden/motion_lib.npz, kin_ops.npz).
"""
import numpy as np
import torch

HINGE, SPHERICAL = 1, 2


# ---- torch_util ------------------------------------------------------------------------------------------------------
def normalize(x, eps=1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)


def quat_mul(a, b):  # the 9-multiply factored product
    shape = a.shape
    a = a.reshape(-1, 4); b = b.reshape(-1, 4)
    x1, y1, z1, w1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x2, y2, z2, w2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([x, y, z, w], dim=-1).view(shape)


def quat_conjugate(q):
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def quat_pos(q):
    z = (q[..., 3:] < 0).float()
    return (1.0 - 2.0 * z) * q


def quat_rotate(q, v):
    qv = q[..., :3]
    t = 2.0 * torch.cross(qv, v, dim=-1)
    return v + q[..., 3:] * t + torch.cross(qv, t, dim=-1)


def quat_to_axis_angle(q):
    q = quat_pos(q)
    length = torch.norm(q[..., :3], dim=-1, p=2)
    angle = 2.0 * torch.atan2(length, q[..., 3])
    axis = q[..., :3] / torch.clamp_min(length, 1e-6).unsqueeze(-1)
    ok = length > 1e-5
    default_axis = torch.zeros_like(axis); default_axis[..., 2] = 1.0
    angle = torch.where(ok, angle, torch.zeros_like(angle))
    axis = torch.where(ok.unsqueeze(-1), axis, default_axis)
    return axis, angle


def axis_angle_to_quat(axis, angle):
    theta = (angle / 2).unsqueeze(-1)
    axis = normalize(axis)
    return normalize(torch.cat([axis * theta.sin(), theta.cos()], dim=-1))


def normalize_angle(x):
    return torch.atan2(torch.sin(x), torch.cos(x))


def exp_map_to_quat(e):
    angle = torch.norm(e, dim=-1)
    axis = e / angle.unsqueeze(-1)
    angle = normalize_angle(angle)
    default_axis = torch.zeros_like(e); default_axis[..., 2] = 1.0
    ok = torch.abs(angle) > 1e-5
    angle = torch.where(ok, angle, torch.zeros_like(angle))
    axis = torch.where(ok.unsqueeze(-1), axis, default_axis)
    return axis_angle_to_quat(axis, angle)


def quat_to_exp_map(q):
    axis, angle = quat_to_axis_angle(q)
    return angle.unsqueeze(-1) * axis


def slerp(q0, q1, t):  # t broadcastable to q0[..., :1]
    cos_half = torch.sum(q0 * q1, dim=-1)
    neg = cos_half < 0
    q1 = torch.where(neg.unsqueeze(-1), -q1, q1)
    cos_half = torch.abs(cos_half).unsqueeze(-1)
    half = torch.acos(cos_half)
    sin_half = torch.sqrt(1.0 - cos_half * cos_half)
    ra = torch.sin((1 - t) * half) / sin_half
    rb = torch.sin(t * half) / sin_half
    new_q = ra * q0 + rb * q1
    new_q = torch.where(torch.abs(sin_half) < 0.001, 0.5 * q0 + 0.5 * q1, new_q)
    new_q = torch.where(torch.abs(cos_half) >= 1, q0, new_q)
    return new_q


# ---- kin_char_model --------------------------------------------------------------------------------------------------
class CharModel:
    def __init__(self, parent, local_translation, local_rotation, joint_type, joint_axis, dof_idx, dof_size):
        self.parent = [int(p) for p in parent]
        self.lt = torch.as_tensor(np.asarray(local_translation, np.float32))
        self.lr = torch.as_tensor(np.asarray(local_rotation, np.float32))
        self.jtype = [int(t) for t in joint_type]
        self.axis = torch.as_tensor(np.asarray(joint_axis, np.float32))
        self.dof_idx = [int(d) for d in dof_idx]
        self.D = int(dof_size)
        self.B = len(self.parent)

    def dof_to_rot(self, dof):
        n = dof.shape[:-1]
        out = torch.zeros(*n, self.B - 1, 4, dtype=dof.dtype)
        out[..., 3] = 1.0
        for j in range(1, self.B):  # python loop over the joints, like the reference
            d = self.dof_idx[j]
            if self.jtype[j] == HINGE:
                out[..., j - 1, :] = axis_angle_to_quat(self.axis[j].expand(*n, 3), dof[..., d])
            elif self.jtype[j] == SPHERICAL:
                out[..., j - 1, :] = exp_map_to_quat(dof[..., d:d + 3])
        return out

    def rot_to_dof(self, jr):
        n = jr.shape[:-2]
        dof = torch.zeros(*n, self.D, dtype=jr.dtype)
        for j in range(1, self.B):
            d = self.dof_idx[j]
            q = jr[..., j - 1, :]
            if self.jtype[j] == HINGE:
                ax, ang = quat_to_axis_angle(q)
                dot = torch.sum(self.axis[j] * ax, dim=-1)
                ang = torch.where(dot < 0, -ang, ang)
                dof[..., d] = ang
            elif self.jtype[j] == SPHERICAL:
                dof[..., d:d + 3] = quat_to_exp_map(q)
        return dof

    def forward_kinematics(self, root_pos, root_rot, joint_rot):
        pos = [root_pos]; rot = [root_rot]
        for j in range(1, self.B):  # python loop over the bodies in DFS order, like the reference
            p = self.parent[j]
            pos.append(pos[p] + quat_rotate(rot[p], self.lt[j].expand_as(root_pos)))
            rot.append(quat_mul(rot[p], quat_mul(self.lr[j].expand_as(root_rot), joint_rot[..., j - 1, :])))
        return torch.stack(pos, dim=-2), torch.stack(rot, dim=-2)


# ---- motion_lib ------------------------------------------------------------------------------------------------------
class MotionLib:
    """Frame tables as the reference keeps them (flat [F, ...] tensors + per-motion metadata)."""

    def __init__(self, tables):
        g = lambda k, dt=torch.float32: torch.as_tensor(np.asarray(tables[k])).to(dt)
        self.root_pos, self.root_rot, self.joint_rot = g("frame_root_pos"), g("frame_root_rot"), g("frame_joint_rot")
        self.root_vel, self.root_ang_vel, self.dof_vel = g("frame_root_vel"), g("frame_root_ang_vel"), g("frame_dof_vel")
        self.contacts = g("frame_contacts")
        self.num_frames = g("motion_num_frames", torch.int64)
        self.lengths = g("motion_lengths")
        self.loop_modes = g("motion_loop_modes", torch.int64)
        self.start_idx = g("motion_start_idx", torch.int64)
        self.root_pos_delta = g("motion_root_pos_delta")

    def calc_frame_blend(self, ids, times):
        length = self.lengths[ids]
        phase = times / length
        wrap = self.loop_modes[ids] == 1
        phase = torch.where(wrap, phase - torch.floor(phase), phase)
        phase = torch.clip(phase, 0.0, 1.0)
        nf = self.num_frames[ids]
        idx0 = (phase * (nf - 1)).long()
        idx1 = torch.min(idx0 + 1, nf - 1)
        blend = phase * (nf - 1) - idx0
        start = self.start_idx[ids]
        return idx0 + start, idx1 + start, blend

    def calc_motion_frame(self, ids, times):
        i0, i1, blend = self.calc_frame_blend(ids, times)
        b = blend.unsqueeze(-1)
        root_pos = (1.0 - b) * self.root_pos[i0] + b * self.root_pos[i1]
        root_rot = slerp(self.root_rot[i0], self.root_rot[i1], b)
        joint_rot = slerp(self.joint_rot[i0], self.joint_rot[i1], b.unsqueeze(-1))
        contacts = (1.0 - b) * self.contacts[i0] + b * self.contacts[i1]
        wrap = self.loop_modes[ids] == 1
        phase = torch.floor(times / self.lengths[ids])
        root_pos = root_pos + torch.where(wrap, phase, torch.zeros_like(phase)).unsqueeze(-1) * self.root_pos_delta[ids]
        return root_pos, root_rot, self.root_vel[i0], self.root_ang_vel[i0], joint_rot, self.dof_vel[i0], contacts


def step_path(cm, lib, ids, times, char_root_pos, char_root_rot, char_dof, dt, tar_steps=(1, 2, 3, 10, 20, 30)):
    """The motion_lib / kin_char_model calls of ONE env step for a batch of envs: reference frame at t and
    at the six look-ahead times (one batched query, as dm_env.py:594 does), FK of character / reference / targets, dof <-> rot."""
    n = ids.shape[0]
    rp, rr, rv, rav, jr, dv, ct = lib.calc_motion_frame(ids, times)                               # _update_ref_motion
    ref_bp, _ = cm.forward_kinematics(rp, rr, jr)
    ref_dof = cm.rot_to_dof(jr)
    steps = torch.tensor(tar_steps, dtype=torch.float32) * dt
    tt = (times.unsqueeze(-1) + steps).reshape(-1)                                               # fetch_tar_obs_data
    tid = ids.unsqueeze(-1).expand(n, len(tar_steps)).reshape(-1)
    trp, trr, _, _, tjr, _, tct = lib.calc_motion_frame(tid, tt)
    tar_bp, _ = cm.forward_kinematics(trp, trr, tjr)
    cjr = cm.dof_to_rot(char_dof)                                                                # obs + reward of the character
    char_bp, _ = cm.forward_kinematics(char_root_pos, char_root_rot, cjr)
    return ref_bp, ref_dof, tar_bp, char_bp, ct, tct


# ---- the whole kinematic step (round 4) --------------------------------------------------------------------------------------------
# IGEnv._post_physics_step (ig_env.py:368-377) as the reference issues it -- batched tensor ops, python loops where the reference has
# them --: height rays, time, reference frame + FK + dof, observation (character | targets | target contacts | character contacts | rays),
# DeepMimic reward + contact term, compute_done + the motion-end rule.  (The sequential fail-rate EMA of dm_env.py:646-660 is a python loop
# over env ids in the reference, not tensor work: not part of this path.)  Default switches of dm_env_default.yaml: heading-frame
# observations, track_root, contact_bodies = [].  Pinned by tests/test_oracle_golden.py::test_torch_full_step_vs_reference_golden against the
# reference's own outputs (tests/golden/env_step.npz).
def quat_diff_angle(q0, q1):   # torch_util.py:454-462
    _, ang = quat_to_axis_angle(quat_mul(q1, quat_conjugate(q0)))
    return ang


def quat_to_tan_norm(q):       # torch_util.py:393-404
    tx = torch.zeros_like(q[..., :3]); tx[..., 0] = 1.0
    nz = torch.zeros_like(q[..., :3]); nz[..., 2] = 1.0
    return torch.cat([quat_rotate(q, tx), quat_rotate(q, nz)], dim=-1)


def calc_heading(q):           # torch_util.py:502-511
    ref = torch.zeros_like(q[..., :3]); ref[..., 0] = 1.0
    d = quat_rotate(q, ref)
    return torch.atan2(d[..., 1], d[..., 0])


def calc_heading_quat_inv(q):  # torch_util.py:523-530
    z = torch.zeros_like(q[..., :3]); z[..., 2] = 1.0
    return axis_angle_to_quat(z, -calc_heading(q))


def rotate_2d_vec(v, angle):   # torch_util.py:651-663
    c, s = torch.cos(angle), torch.sin(angle)
    return torch.stack([v[..., 0] * c - v[..., 1] * s, v[..., 0] * s + v[..., 1] * c], dim=-1)


class Terrain:
    """SubTerrain (terrain_util.py:18-30): hf[X][Y], nearest cell with torch.round (:146-156), indices clamped."""

    def __init__(self, hf, min_point, dxdy):
        self.hf = torch.as_tensor(np.asarray(hf, np.float32))
        self.min_point = torch.as_tensor(np.asarray(min_point, np.float32))
        self.dxdy = torch.as_tensor(np.asarray(dxdy, np.float32))

    def hf_vals(self, xy):
        idx = torch.round((xy - self.min_point) / self.dxdy).long()
        ix = torch.clamp(idx[..., 0], 0, self.hf.shape[0] - 1)
        iy = torch.clamp(idx[..., 1], 0, self.hf.shape[1] - 1)
        return self.hf[ix, iy]


class KinematicStep:
    """State + constants of N envs; `step()` = one `_post_physics_step` on the state the caller injected."""

    def __init__(self, cm, lib, terrain, cfg):
        self.cm, self.lib, self.ter = cm, lib, terrain
        T = lambda x, dt=torch.float32: torch.as_tensor(np.asarray(x)).to(dt)
        self.key_ids = [int(k) for k in cfg["key_body_ids"]]
        self.tar_steps = T(cfg["tar_obs_steps"])
        self.ray_points = T(cfg["ray_points"])
        self.env_offsets = T(cfg["env_offsets"])
        self.motion_offsets = T(cfg["motion_offsets"])           # [M][T][2]
        self.dt = float(cfg["timestep"])
        self.episode_length = float(cfg["episode_length"])
        self.min_h, self.max_h = float(cfg["min_obs_h"]), float(cfg["max_obs_h"])
        self.w = [float(x) for x in cfg["reward_weights"]]        # pose, vel, root_pos, root_vel, key_pos (already normalised)
        self.joint_err_w, self.dof_err_w = T(cfg["joint_err_w"]), T(cfg["dof_err_w"])
        self.contact_w = T(cfg["contact_weights"])
        self.pose_term_dist = T(cfg["pose_termination_dist"])
        self.root_pos_term, self.root_rot_term = float(cfg["root_pos_termination_dist"]), float(cfg["root_rot_termination_angle"])

    def _to_terrain(self, xy, st):   # dm_env.py:554-558
        off = self.motion_offsets[st["motion_ids"], st["terrain_ids"]] - self.env_offsets[:, 0:2]
        return xy + off

    def step(self, st):
        cm, lib = self.cm, self.lib
        n = st["char_root_pos"].shape[0]
        B, J = cm.B, cm.B - 1
        root_pos, root_rot = st["char_root_pos"], st["char_root_rot"]
        # ---- height rays (ig_parkour_env.py:_refresh_ray_obs_hfs, mgdm_dm_util.py:128-145)
        heading = calc_heading(root_rot)
        g = root_pos + self.env_offsets
        pts = rotate_2d_vec(self.ray_points.unsqueeze(0), heading.unsqueeze(-1)) + g[:, None, 0:2]
        hf = torch.clamp(self.ter.hf_vals(pts) - g[:, None, 2], self.min_h, self.max_h)
        # ---- time (ig_env.py:391-394) and reference motion (dm_env.py:523-545)
        st["timestep"] = st["timestep"] + 1
        time = self.dt * st["timestep"].float()
        mt = time + st["time_offsets"]
        rp, rr, rv, rav, jr, dv, ct = lib.calc_motion_frame(st["motion_ids"], mt)
        rp = torch.cat([self._to_terrain(rp[:, 0:2], st), rp[:, 2:3]], dim=-1)
        ref_bp, _ = cm.forward_kinematics(rp, rr, jr)
        # ---- observation (ig_parkour_env.py:842-965; compute_char_obs ig_char_env.py:582-627; compute_tar_obs mgdm_dm_util.py:405-460)
        cjr = cm.dof_to_rot(st["char_dof_pos"])
        body_pos, _ = cm.forward_kinematics(root_pos, root_rot, cjr)
        hinv = calc_heading_quat_inv(root_rot)
        key = body_pos[:, self.key_ids, :] - root_pos.unsqueeze(1)
        char_obs = torch.cat([quat_to_tan_norm(quat_mul(hinv, root_rot)), quat_rotate(hinv, st["char_root_vel"]), quat_rotate(hinv, st["char_root_ang_vel"]),
                              quat_to_tan_norm(cjr).reshape(n, 6 * J), st["char_dof_vel"],
                              quat_rotate(hinv.unsqueeze(1).expand(n, len(self.key_ids), 4), key).reshape(n, -1)], dim=-1)
        S = self.tar_steps.shape[0]
        tt = (mt.unsqueeze(-1) + self.dt * self.tar_steps).reshape(-1)
        tid = st["motion_ids"].unsqueeze(-1).expand(n, S).reshape(-1)
        trp, trr, _, _, tjr, _, tct = lib.calc_motion_frame(tid, tt)
        off = (self.motion_offsets[st["motion_ids"], st["terrain_ids"]] - self.env_offsets[:, 0:2]).unsqueeze(1).expand(n, S, 2).reshape(-1, 2)
        trp = torch.cat([trp[:, 0:2] + off, trp[:, 2:3]], dim=-1)
        tbp, _ = cm.forward_kinematics(trp, trr, tjr)
        hinv_s = hinv.unsqueeze(1).expand(n, S, 4).reshape(-1, 4)
        root_s = root_pos.unsqueeze(1).expand(n, S, 3).reshape(-1, 3)
        rpl = quat_rotate(hinv_s, trp - root_s)
        tkey = tbp[:, self.key_ids, :] - trp.unsqueeze(1)
        tkey = quat_rotate(hinv_s.unsqueeze(1).expand(-1, len(self.key_ids), 4), tkey) + rpl.unsqueeze(1)
        tar_obs = torch.cat([rpl, quat_to_tan_norm(quat_mul(hinv_s, trr)), quat_to_tan_norm(tjr).reshape(n * S, 6 * J), tkey.reshape(n * S, -1)], dim=-1)
        cf = st["contact_forces"]
        char_contacts = (torch.norm(cf, dim=-1) > 1e-5).float()
        obs = torch.cat([char_obs, tar_obs.reshape(n, -1), tct.reshape(n, S * B), char_contacts, hf], dim=-1)
        # ---- reward (mgdm_dm_util.py:270-333, contact term :498-518, ig_parkour_env.py:984-1046); simulator rigid-body positions for the keys
        sim_bp = st["char_body_pos"]
        d = quat_diff_angle(cjr, jr)
        pose_err = torch.sum(self.joint_err_w * d * d, dim=-1)
        v = dv - st["char_dof_vel"]
        vel_err = torch.sum(self.dof_err_w * v * v, dim=-1)
        rpd = rp - root_pos
        root_pos_err = torch.sum(rpd * rpd, dim=-1)
        ra = quat_diff_angle(root_rot, rr)
        root_rot_err = ra * ra
        dvr = rv - st["char_root_vel"]; dva = rav - st["char_root_ang_vel"]
        root_vel_err = torch.sum(dvr * dvr, dim=-1); root_ang_vel_err = torch.sum(dva * dva, dim=-1)
        k0 = sim_bp[:, self.key_ids, :] - root_pos.unsqueeze(1)
        k1 = ref_bp[:, self.key_ids, :] - rp.unsqueeze(1)
        kd = k1 - k0
        key_err = torch.sum(torch.sum(kd * kd, dim=-1), dim=-1)
        pose_r = torch.exp(-0.25 * pose_err); vel_r = torch.exp(-0.01 * vel_err)
        root_pose_r = torch.exp(-5.0 * (root_pos_err + 0.1 * root_rot_err)); root_vel_r = torch.exp(-1.0 * (root_vel_err + 0.1 * root_ang_vel_err))
        key_r = torch.exp(-10.0 * key_err)
        rew = self.w[0] * pose_r + self.w[1] * vel_r + self.w[2] * root_pose_r + self.w[3] * root_vel_r + self.w[4] * key_r
        fn = torch.clamp_max(torch.norm(cf, dim=-1), 1.0)
        cr = self.contact_w * (-(1.0 - ct) * fn + ct * fn)
        rew = rew + torch.mean(cr, dim=-1)
        # ---- done (mgdm_dm_util.py:335-402 with contact_bodies = []; dm_env.py:628-665 motion end => FAIL)
        done = torch.zeros(n, dtype=torch.int32)
        done[time >= self.episode_length] = 2                                     # DoneFlags.TIME
        bp_rel = sim_bp[:, 1:, :] - sim_bp[:, 0:1, :]
        tp_rel = ref_bp[:, 1:, :] - ref_bp[:, 0:1, :]
        dd = tp_rel - bp_rel
        pose_fail = torch.any(torch.sum(dd * dd, dim=-1) > self.pose_term_dist * self.pose_term_dist, dim=-1)
        r0 = sim_bp[:, 0, :] - ref_bp[:, 0, :]
        pose_fail = pose_fail | (torch.sum(r0 * r0, dim=-1) > self.root_pos_term * self.root_pos_term)
        pose_fail = pose_fail | (torch.abs(quat_diff_angle(root_rot, rr)) > self.root_rot_term)
        failed = pose_fail & (time > 1e-5)
        done[failed] = 1                                                          # DoneFlags.FAIL
        motion_end = (mt >= lib.lengths[st["motion_ids"]]) & (lib.loop_modes[st["motion_ids"]] != 1)
        done[motion_end] = 1
        return obs, rew, done


def make_tables(clips, frame_root_vel, frame_root_ang_vel, frame_dof_vel, num_bodies=15):
    """Flat frame tables for MotionLib from the clips (dicts with root_pos / root_rot / joint_rot / contacts / fps / loop_mode) and the
    load-time velocity tables (MotionLib._load_motion_file :318-327; the tests and bench.py take them from the CPU oracle's library)."""
    nf = np.array([c["root_pos"].shape[0] for c in clips], np.int64)
    return dict(frame_root_pos=np.concatenate([c["root_pos"] for c in clips]), frame_root_rot=np.concatenate([c["root_rot"] for c in clips]),
                frame_joint_rot=np.concatenate([c["joint_rot"] for c in clips]), frame_root_vel=frame_root_vel, frame_root_ang_vel=frame_root_ang_vel,
                frame_dof_vel=frame_dof_vel,
                frame_contacts=np.concatenate([c["contacts"] if c["contacts"] is not None else np.zeros((c["root_pos"].shape[0], num_bodies), np.float32) for c in clips]),
                motion_num_frames=nf, motion_lengths=np.array([(c["root_pos"].shape[0] - 1) / c["fps"] for c in clips], np.float32),
                motion_loop_modes=np.array([c["loop_mode"] for c in clips], np.int64), motion_start_idx=np.concatenate([[0], np.cumsum(nf)[:-1]]),
                motion_root_pos_delta=np.stack([np.append(c["root_pos"][-1, :2] - c["root_pos"][0, :2], 0.0) for c in clips]).astype(np.float32))
