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
