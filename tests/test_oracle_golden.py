"""Pin the CPU oracle (oracle/parc_oracle.c) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU-only; tolerance 2e-6 unless a case states otherwise."""
import os

import numpy as np
import pytest

from conftest import golden

TOL = 2e-6


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    # ABSOLUTE error against tol + 2 ulp of the value (fp32: 2.4e-7 |b|) -- round 3 used a tolerance RELATIVE to max(1, |b|), i.e. 5e-4 m at 50 m;
    # the ulp term is what any fp32 evaluation order may differ by at that magnitude (the far-origin fixture measures it), tol = 0 stays exact
    err = np.abs(a - b) / (1.0 + (2.4e-7 / tol) * np.abs(b)) if tol > 0 else np.abs(a - b)
    err[both_nan] = 0.0
    assert np.nanmax(err) <= tol and not np.any(np.isnan(err)), f"{what}: max err {np.nanmax(err)} at {np.nanargmax(err)}"


def test_quat_ops(oracle):
    g = golden("quat_ops")
    a, b, v, t, e, axis, angle = (g[k] for k in ["a", "b", "v", "t", "e", "axis", "angle"])
    close(oracle.quat_mul(a, b), g["quat_mul"], what="quat_mul")
    close(oracle.quat_multiply(a, b), g["quat_multiply"], what="quat_multiply")
    close(oracle.quat_rotate(a, v), g["quat_rotate"], what="quat_rotate")
    close(oracle.quat_conjugate(a), g["quat_conjugate"])
    close(oracle.quat_pos(a), g["quat_pos"])
    close(oracle.normalize(v), g["normalize"])
    ax, an = oracle.quat_to_axis_angle(a)
    close(ax, g["q2aa_axis"], what="q2aa axis"); close(an, g["q2aa_angle"], what="q2aa angle")
    close(oracle.axis_angle_to_quat(axis, angle), g["aa2q"], what="aa2q")
    eax, ean = oracle.exp_map_to_axis_angle(e)
    close(eax, g["e2aa_axis"], what="e2aa axis"); close(ean, g["e2aa_angle"], what="e2aa angle")
    close(oracle.exp_map_to_quat(e), g["exp_map_to_quat"], what="exp_map_to_quat")
    close(oracle.quat_to_exp_map(a), g["quat_to_exp_map"], what="quat_to_exp_map")
    close(oracle.quat_diff(a, b), g["quat_diff"])
    # near-opposite/identical pairs amplify 1-ulp differences of the product through atan2 near 0/pi
    close(oracle.quat_diff_angle(a, b), g["quat_diff_angle"], tol=2e-4, what="quat_diff_angle")
    close(oracle.quat_normalize(a * np.float32(1.7)), g["quat_normalize"])
    close(oracle.quat_to_tan_norm(a), g["quat_to_tan_norm"])
    close(oracle.slerp(a, b, t), g["slerp"], tol=5e-6, what="slerp")
    close(oracle.calc_heading(a), g["calc_heading"], tol=5e-6)
    close(oracle.calc_heading_quat_inv(a), g["calc_heading_quat_inv"])
    close(oracle.rotate_2d_vec(v[:, :2], angle), g["rotate_2d_vec"])
    close(oracle.normalize_angle(angle * np.float32(3.0)), g["normalize_angle"], tol=5e-6)


def test_kin_ops(oracle, orc_char):
    g = golden("kin_ops")
    close(oracle.dof_to_rot(orc_char, g["dof"]), g["joint_rot"], what="dof_to_rot")
    close(oracle.rot_to_dof(orc_char, g["joint_rot"]), g["dof_back"], tol=5e-6, what="rot_to_dof")
    close(oracle.rot_to_dof(orc_char, g["joint_rot_rand"]), g["dof_rand"], tol=5e-6, what="rot_to_dof rand")
    bp, br = oracle.forward_kinematics(orc_char, g["root_pos"], g["root_rot"], g["joint_rot"])
    close(bp, g["body_pos"], what="fk pos"); close(br, g["body_rot"], what="fk rot")
    bp, br = oracle.forward_kinematics(orc_char, g["root_pos"], g["root_rot"], g["joint_rot_rand"])
    close(bp, g["body_pos_rand"]); close(br, g["body_rot_rand"])
    close(oracle.compute_dof_vel(orc_char, g["joint_rot"], g["joint_rot1"], 1.0 / 30.0), g["dof_vel"], tol=2e-5, what="dof_vel")


def test_known_answers_survey(oracle, orc_char):
    """SURVEY.md §8(c) known-answer samples on sfu.pkl (measured on the reference during the survey)."""
    from helpers import load_clips, make_orc_mlib
    clips = load_clips(["sfu"])
    lib = make_orc_mlib(oracle, orc_char, clips, [1.0])
    i0, i1, bl = oracle.mlib_calc_frame_blend(lib, [0, 0], [0.10, 0.21])
    assert (i0[0], i1[0]) == (3, 4) and abs(bl[0]) < 1e-6
    assert (i0[1], i1[1]) == (6, 7) and abs(bl[1] - 0.2999997) < 1e-6
    o = oracle.mlib_calc_motion_frame(lib, [0, 0], [0.10, 0.21])
    close(o["root_pos"][0], [10.8036976, 1.8566505, 0.7566922], tol=1e-6)
    close(o["root_pos"][1], [10.8226118, 2.3026748, 0.7736420], tol=1e-6)
    close(o["root_rot"][1], [-0.0113259, -0.0050410, 0.6899159, 0.7237490], tol=1e-6)
    bp, _ = oracle.forward_kinematics(orc_char, o["root_pos"], o["root_rot"], o["joint_rot"])
    close(bp[0, 11], [10.9086103, 1.8863832, 0.1450104], tol=1e-6)
    close(bp[0, 8], [10.5566044, 2.0967307, 0.7700925], tol=1e-6)
    close(bp[1, 11], [10.9210749, 2.3483460, 0.0427647], tol=1e-6)
    close(bp[1, 8], [10.5470676, 2.4380567, 0.7927424], tol=1e-6)
    dof = oracle.rot_to_dof(orc_char, o["joint_rot"])
    close(dof[1, 0:6], [-0.0387498, 0.4994673, -0.0135002, 0.0099086, 0.1129964, 0.0633409], tol=1e-6)
    close(o["dof_vel"][1, 0:4], [-1.5708344, 0.0133573, 1.3118259, 0.2720484], tol=2e-6)
    ray = oracle.ray_points_cone(0.05, 2, 60, 3, 3, 0.26179938779)
    close(ray[0], [-0.0707107, 0.0707107], tol=1e-6); close(ray[62], [2.1213202, -2.1213202], tol=1e-6)
    close(ray[3 * 63 + 62], [3.0, 0.0], tol=1e-6); close(ray[6 * 63 + 62], [2.1213202, 2.1213202], tol=1e-6)


def test_motion_lib(oracle, orc_char):
    from helpers import CLIPS4, load_clips, make_orc_mlib
    g = golden("motion_lib")
    clips = load_clips(CLIPS4)
    lib = make_orc_mlib(oracle, orc_char, clips, [1.0, 1.5, 2.0, 2.5])
    F = int(g["frame_root_pos"].shape[0]); M = 4
    A = oracle.mlib_array
    close(A(lib, "motion_weights", (M,)), g["motion_weights"])
    close(A(lib, "motion_lengths", (M,)), g["motion_lengths"], tol=0)
    close(A(lib, "motion_dt", (M,)), g["motion_dt"], tol=0)
    assert np.array_equal(A(lib, "motion_start_idx", (M,), np.int64), g["motion_start_idx"])
    close(A(lib, "motion_root_pos_delta", (M, 3)), g["motion_root_pos_delta"], tol=0)
    close(A(lib, "frame_root_vel", (F, 3)), g["frame_root_vel"], tol=0, what="root_vel")
    close(A(lib, "frame_root_ang_vel", (F, 3)), g["frame_root_ang_vel"], tol=3e-5, what="root_ang_vel")
    close(A(lib, "frame_dof_vel", (F, 28)), g["frame_dof_vel"], tol=3e-5, what="dof_vel")
    i0, i1, bl = oracle.mlib_calc_frame_blend(lib, g["q_ids"], g["q_times"])
    assert np.array_equal(i0, g["idx0"]) and np.array_equal(i1, g["idx1"])
    close(bl, g["blend"], tol=0, what="blend")
    o = oracle.mlib_calc_motion_frame(lib, g["q_ids"], g["q_times"])
    for k in ["root_pos", "root_rot", "joint_rot", "contacts"]:
        close(o[k], g[k], tol=5e-6, what=k)
    for k in ["root_vel", "root_ang_vel", "dof_vel"]:
        close(o[k], g[k], tol=3e-5, what=k)
    bp, _ = oracle.forward_kinematics(orc_char, o["root_pos"], o["root_rot"], o["joint_rot"])
    close(bp, g["body_pos"], tol=5e-6)
    close(oracle.rot_to_dof(orc_char, o["joint_rot"]), g["dof_pos"], tol=1e-5)


def test_motion_lib_wrap(oracle, orc_char):
    from helpers import load_clips, make_orc_mlib
    g = golden("motion_lib_wrap")
    clips = load_clips(["civilization"])
    clips[0]["loop_mode"] = 1
    lib = make_orc_mlib(oracle, orc_char, clips, [1.0])
    o = oracle.mlib_calc_motion_frame(lib, np.zeros(64, np.int64), g["q_times"])
    for k in ["root_pos", "root_rot", "joint_rot", "contacts"]:
        close(o[k], g[k], tol=5e-6, what=k)


def test_terrain_lookup(oracle):
    g = golden("terrain_lookup")
    close(oracle.ray_points_cone(0.05, 2, 60, 3, 3, 0.26179938779), g["ray_points"], tol=1e-6, what="ray fan")
    t = oracle.make_terrain(g["hf"], g["min_point"], g["dxdy"])
    assert np.array_equal(oracle.terrain_grid_index(t, g["points"]), g["grid_index"])
    assert np.array_equal(oracle.terrain_hf_vals(t, g["points"]), g["hf_vals"])


def test_done_and_contact_reward(oracle):
    from helpers import default_cfg
    g = golden("done_table")
    cfg = default_cfg(oracle, 128)
    d = oracle.compute_done(cfg, 15, g["time"], g["root_rot"], g["body_pos"], g["tar_root_rot"], g["tar_body_pos"])
    assert np.array_equal(d, g["done"])
    assert set(np.unique(d)) >= {0, 1}
    close(oracle.contact_reward(g["tar_contacts"], g["contact_forces"], np.full(15, 5.0, np.float32)), g["contact_r"])


def test_env_step_pipeline(oracle, orc_char):
    """The reference's own IGEnv._post_physics_step on injected state, 3 consecutive control steps."""
    from helpers import build_oracle_scene, load_state_into
    g = golden("env_step")
    scene = build_oracle_scene(oracle, orc_char, g)
    st = scene["state"]
    for s in range(3):
        load_state_into(st, g, f"s{s}_in_")
        oracle.env_post_physics_step(orc_char, scene["lib"], scene["terrain"], scene["cfg"], st)
        oracle.env_update_curriculum(scene["lib"], scene["cfg"], st)
        p = f"s{s}_out_"
        assert np.array_equal(st["timestep_buf"], g[p + "timestep"])
        close(st["time_buf"], g[p + "time"], tol=0, what="time")
        for k in ["ref_root_pos", "ref_root_rot", "ref_joint_rot", "ref_body_pos", "ref_contacts", "ref_dof_pos"]:
            close(st[k], g[p + k], tol=1e-5, what=k)
        for k in ["ref_root_vel", "ref_root_ang_vel", "ref_dof_vel"]:
            close(st[k], g[p + k], tol=3e-5, what=k)
        # rays: identical except where a 1-ulp sin/cos difference moves a point across a cell edge
        ray_bad = np.abs(st["ray_hfs"] - g[p + "ray_hfs"]) > 1e-5
        assert ray_bad.mean() < 2e-4, ray_bad.sum()
        obs_err = np.abs(st["obs"] - g[p + "obs"])
        obs_err[:, 871:][ray_bad] = 0
        assert obs_err.max() <= 1e-5, (obs_err.max(), np.unravel_index(obs_err.argmax(), obs_err.shape))
        close(st["reward"], g[p + "reward"], tol=1e-5, what="reward")
        names = ["pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "total_r"]
        for i, nm in enumerate(names):
            close(st["reward_terms"][i], g[p + "r_" + nm], tol=1e-5, what=nm)
        close(st["tracking_error"], g[p + "tracking_error"], tol=1e-5, what="tracking_error")
        assert np.array_equal(st["done"], g[p + "done"])
        close(st["fail_rates"], g[p + "fail_rates"], tol=0, what="fail_rates")


def test_env_reset(oracle, orc_char):
    from helpers import build_oracle_scene
    g = golden("env_step")
    scene = build_oracle_scene(oracle, orc_char, g)
    st = scene["state"]
    n = scene["cfg"].num_envs
    ids = np.arange(n)
    oracle.env_reset_with(orc_char, scene["lib"], scene["terrain"], scene["cfg"], st, ids, g["reset_motion_ids"],
                          g["reset_terrain_ids"], g["reset_time_offsets"], g["reset_xy_noise"])
    for k in ["char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos", "char_dof_vel"]:
        close(st[k], g["reset_" + k], tol=3e-5, what=k)
    close(st["char_root_pos"], g["reset_char_root_pos"], tol=1e-5)
    close(st["ref_root_pos"], g["reset_ref_root_pos"], tol=1e-5)
    close(st["ref_contacts"], g["reset_ref_contacts"], tol=1e-5)
    oracle.env_refresh_rays(scene["terrain"], scene["cfg"], st)
    oracle.env_compute_obs(orc_char, scene["lib"], scene["cfg"], st, ids)
    ray_bad = np.abs(st["ray_hfs"] - g["reset_ray_hfs"]) > 1e-5
    assert ray_bad.mean() < 2e-4
    err = np.abs(st["obs"] - g["reset_obs"]); err[:, 871:][ray_bad] = 0
    assert err.max() <= 1e-5, err.max()


def test_recorded_isaacgym_obs(oracle, orc_char):
    """an obs stream recorded from a real Isaac Gym run ships in
    dec2024_teaser_717_1_opt_dm.pkl.  Root tan-norm, 10 of 14 joint tan-norms (the 4 arm joints used an
    older XML convention), contact flags and all in-bounds height rays must reproduce from the file's frames."""
    from helpers import load_clips
    g = golden("recorded_obs_dec2024_teaser_717_1_opt_dm")
    obs = g["obs"]
    clip = load_clips(["dec2024_teaser_717_1_opt_dm"])[0]
    n = obs.shape[0]
    assert obs.shape == (142, 1312) and clip["root_pos"].shape[0] == n
    hinv = oracle.calc_heading_quat_inv(clip["root_rot"])
    local = oracle.quat_mul(hinv, clip["root_rot"])
    close(oracle.quat_to_tan_norm(local), obs[:, 0:6], tol=5e-6, what="root tan-norm")
    jt = oracle.quat_to_tan_norm(clip["joint_rot"].reshape(-1, 4)).reshape(n, 14, 6)
    rec = obs[:, 12:96].reshape(n, 14, 6)
    good = [j for j in range(14) if j not in (2, 3, 5, 6)]  # arm joints: older convention in the recording
    close(jt[:, good], rec[:, good], tol=5e-6, what="joint tan-norm")
    assert np.array_equal(obs[:, 856:871], clip["contacts"])
    # height rays on the clip's own terrain slice: exact on every ray whose sample lies inside the slice
    t = oracle.make_terrain(clip["hf"], clip["min_point"], [clip["dx"], clip["dx"]])
    ray = golden("terrain_lookup")["ray_points"]
    heading = oracle.calc_heading(clip["root_rot"])
    X, Y = clip["hf"].shape
    tot = 0; ok = 0
    for i in range(n):
        p = oracle.rotate_2d_vec(ray, np.full(441, heading[i], np.float32)) + clip["root_pos"][i, 0:2]
        idx_f = (p - clip["min_point"]) / np.float32(clip["dx"])
        inb = (idx_f[:, 0] > 0.01) & (idx_f[:, 0] < X - 1.01) & (idx_f[:, 1] > 0.01) & (idx_f[:, 1] < Y - 1.01)
        h = np.clip(oracle.terrain_hf_vals(t, p) - clip["root_pos"][i, 2], -3.0, 3.0)
        d = np.abs(h - obs[i, 871:])[inb]
        tot += inb.sum(); ok += (d < 1e-5).sum()
    assert tot > 0.7 * n * 441 and ok / tot > 0.999, (tot, ok)


def test_torch_path_vs_golden(char_golden):
    """oracle/torch_path.py (the PyTorch-CPU op sequence bench.py times as the north star's CPU baseline) against the golden
    vectors of the real reference."""
    import torch
    from oracle import torch_path as tp
    cg = char_golden
    cm = tp.CharModel(cg["parent"], cg["local_translation"], cg["local_rotation"], cg["joint_type"], cg["joint_axis"], cg["dof_idx"], int(cg["dof_size"]))
    T = lambda a: torch.as_tensor(np.asarray(a))
    k = golden("kin_ops")
    close(cm.dof_to_rot(T(k["dof"])).numpy(), k["joint_rot"], what="dof_to_rot")
    close(cm.rot_to_dof(T(k["joint_rot"])).numpy(), k["dof_back"], tol=5e-6, what="rot_to_dof")
    close(cm.rot_to_dof(T(k["joint_rot_rand"])).numpy(), k["dof_rand"], tol=5e-6, what="rot_to_dof rand")
    bp, br = cm.forward_kinematics(T(k["root_pos"]), T(k["root_rot"]), T(k["joint_rot"]))
    close(bp.numpy(), k["body_pos"], what="fk pos"); close(br.numpy(), k["body_rot"], what="fk rot")
    g = golden("motion_lib")
    lib = tp.MotionLib(g)
    ids, times = T(g["q_ids"]).long(), T(g["q_times"]).float()
    i0, i1, bl = lib.calc_frame_blend(ids, times)
    assert np.array_equal(i0.numpy(), g["idx0"]) and np.array_equal(i1.numpy(), g["idx1"])
    close(bl.numpy(), g["blend"], what="blend")
    out = lib.calc_motion_frame(ids, times)
    for o, name in zip(out, ["root_pos", "root_rot", "root_vel", "root_ang_vel", "joint_rot", "dof_vel", "contacts"]):
        close(o.numpy(), g[name], tol=5e-6, what=name)
    bp, _ = cm.forward_kinematics(out[0], out[1], out[4])
    close(bp.numpy(), g["body_pos"], tol=5e-6, what="body_pos")
    close(cm.rot_to_dof(out[4]).numpy(), g["dof_pos"], tol=5e-6, what="dof_pos")
    w = golden("motion_lib_wrap")  # WRAP loop offset
    import copy
    lib2 = tp.MotionLib({**{kk: g[kk] for kk in g.files}, "motion_loop_modes": np.ones(4, np.int64)})
    # the wrap golden is the civilization clip alone: motion 1 of the 4-clip tables
    o2 = lib2.calc_motion_frame(torch.ones(64, dtype=torch.long), T(w["q_times"]).float())
    close(o2[0].numpy(), w["root_pos"], tol=5e-6, what="wrap root_pos"); close(o2[4].numpy(), w["joint_rot"], tol=5e-6, what="wrap joint_rot")
    res = tp.step_path(cm, lib, ids[:64], times[:64].clamp(min=0.0), T(k["root_pos"][:64]), T(k["root_rot"][:64]), T(k["dof"][:64]), 1.0 / 30.0)
    assert res[2].shape == (64 * 6, 15, 3) and all(torch.isfinite(r).all() for r in res)


def test_fall_termination_with_contact_bodies_vs_reference(oracle, orc_char):
    """`contact_bodies: [right_foot, left_foot]` (not the default config): the fall rule of compute_done (mgdm_dm_util.py:349-360) with
    the per-body terrain lookup of RefCharEnv.update_done (:147-152), against the reference's own `_post_physics_step` on the four
    combinations of (contact on a non-contact body, a non-contact body below the termination height); pose termination off in
    the fixture so that only the fall rule, the time limit and the motion end decide."""
    from helpers import build_oracle_scene, default_cfg, load_state_into
    g = golden("env_step_fall")
    g0 = golden("env_step")
    sc = build_oracle_scene(oracle, orc_char, g0)
    n = g0["env_offsets"].shape[0]
    cfg = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"], pose_termination=bool(g["pose_termination"]),
                      contact_body_ids=[int(b) for b in g["contact_body_ids"]], termination_height=float(g["termination_height"]))
    st = sc["state"]
    load_state_into(st, g, "in_")
    oracle.env_post_physics_step(orc_char, sc["lib"], sc["terrain"], cfg, st)
    oracle.env_update_curriculum(sc["lib"], cfg, st)
    assert np.array_equal(st["done"], g["out_done"])
    assert (g["out_done"][3::4] == 1).all() and (g["out_done"] == 0).sum() > 30     # the fixture exercises both outcomes
    np.testing.assert_allclose(st["obs"], g["out_obs"], atol=1e-5)
    np.testing.assert_array_equal(st["fail_rates"], g["out_fail_rates"])
    # counterfactual: with contact_bodies = [] nobody falls
    cfg0 = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"], pose_termination=False)
    load_state_into(st, g, "in_")
    oracle.env_post_physics_step(orc_char, sc["lib"], sc["terrain"], cfg0, st)
    oracle.env_update_curriculum(sc["lib"], cfg0, st)
    assert (st["done"] != g["out_done"]).sum() >= 8      # (some of the sixteen fall rows also reach their motion end)


def test_env_step_without_root_tracking_vs_reference_golden(oracle, orc_char):
    """`track_root: False` (off the default config): the reward drops the horizontal root position error and compares root rotation, root
    velocities and key positions in each character's OWN heading frame (convert_to_local, mgdm_dm_util.py:247-267, :294-310);
    compute_done skips the root position / rotation termination (:386).  env_step_local_root.npz is the reference's own
    `_post_physics_step` on a state where sixteen characters are turned by up to 2.5 rad and sixteen displaced by metres."""
    from helpers import build_oracle_scene, default_cfg, load_state_into
    g = golden("env_step_local_root")
    g0 = golden("env_step")
    sc = build_oracle_scene(oracle, orc_char, g0)
    n = g0["env_offsets"].shape[0]
    st = sc["state"]
    out = {}
    for track in (False, True):
        cfg = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"], track_root=track)
        load_state_into(st, g, "in_")
        oracle.env_post_physics_step(orc_char, sc["lib"], sc["terrain"], cfg, st)
        oracle.env_update_curriculum(sc["lib"], cfg, st)
        out[track] = (st["reward"].copy(), st["done"].copy())
        if not track:
            assert np.array_equal(st["done"], g["out_done"])
            np.testing.assert_allclose(st["reward"], g["out_reward"], atol=1e-5)
            for k_o, k_g in [("r_root_pos_r", "out_r_root_pos_r"), ("r_root_vel_r", "out_r_root_vel_r"), ("r_key_pos_r", "out_r_key_pos_r")]:
                if k_o in st:
                    np.testing.assert_allclose(st[k_o], g[k_g], atol=1e-5)
            np.testing.assert_allclose(st["obs"], g["out_obs"], atol=1e-5)
            np.testing.assert_array_equal(st["fail_rates"], g["out_fail_rates"])
    # the switch matters on this state: rewards of the turned / displaced rows differ, and with root tracking more episodes end
    assert np.abs(out[True][0][16:40] - out[False][0][16:40]).max() > 0.05
    assert (out[True][1] != 0).sum() > (out[False][1] != 0).sum()


def test_env_step_reward_and_done_switches_vs_reference_golden(oracle, orc_char):
    """`track_root_h: False` (the vertical root error leaves the root-position reward term) and `enable_early_termination: False`
    (compute_done: only the time limit and the motion end finish an episode), each against the reference's own `_post_physics_step`
    (env_step_reward_done_switches.npz: the turned / displaced rows of the local-root fixture + sixteen rows lifted or lowered by up to 0.5 m)."""
    from helpers import build_oracle_scene, default_cfg, load_state_into
    g = golden("env_step_reward_done_switches")
    g0 = golden("env_step")
    sc = build_oracle_scene(oracle, orc_char, g0)
    n = g0["env_offsets"].shape[0]
    st = sc["state"]
    for tag, kw in (("h0_", dict(track_root_h=False)), ("et0_", dict(enable_early_termination=False))):
        res = {}
        for on in (False, True):
            cfg = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"], **({} if on else kw))
            load_state_into(st, g, tag + "in_")
            oracle.env_post_physics_step(orc_char, sc["lib"], sc["terrain"], cfg, st)
            oracle.env_update_curriculum(sc["lib"], cfg, st)
            res[on] = (st["reward"].copy(), st["done"].copy())
            if not on:
                assert np.array_equal(st["done"], g[tag + "out_done"])
                np.testing.assert_allclose(st["reward"], g[tag + "out_reward"], atol=1e-5)
                np.testing.assert_allclose(st["obs"], g[tag + "out_obs"], atol=1e-5)
                np.testing.assert_array_equal(st["fail_rates"], g[tag + "out_fail_rates"])
        # each switch matters on this state: the lifted rows' reward / the number of finished episodes
        if tag == "h0_":
            assert np.abs(res[True][0][40:56] - res[False][0][40:56]).max() > 0.02
        else:
            assert (res[True][1] != 0).sum() > (res[False][1] != 0).sum() + 8


def test_env_step_with_global_observations_vs_reference_golden(oracle, orc_char):
    """`global_obs: True` (off the default config): compute_char_obs (ig_char_env.py:586-589, :603) and compute_tar_obs
    (mgdm_dm_util.py:417) leave root rotation, root velocities, root / key offsets in the global frame, and the targets' key offsets are not
    shifted by the root offset.  env_step_global_obs.npz is the reference's own `_post_physics_step` with that switch."""
    from helpers import build_oracle_scene, default_cfg, load_state_into
    g = golden("env_step_global_obs")
    g0 = golden("env_step")
    sc = build_oracle_scene(oracle, orc_char, g0)
    n = g0["env_offsets"].shape[0]
    st = sc["state"]
    obs = {}
    for gl in (True, False):
        cfg = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"], global_obs=gl)
        load_state_into(st, g, "in_")
        oracle.env_post_physics_step(orc_char, sc["lib"], sc["terrain"], cfg, st)
        oracle.env_update_curriculum(sc["lib"], cfg, st)
        obs[gl] = st["obs"].copy()
        if gl:
            np.testing.assert_allclose(st["obs"], g["out_obs"], atol=1e-5)
            np.testing.assert_allclose(st["reward"], g["out_reward"], atol=1e-5)
            assert np.array_equal(st["done"], g["out_done"])
    d = np.abs(obs[True] - obs[False])
    assert d[:, :12].max() > 0.1 and d[:, 136:766].max() > 0.1      # the switch changes the root / target blocks ...
    assert d[:, 12:124].max() == 0 and d[:, 766:].max() == 0         # ... and nothing else: joint rotations, dof velocities, contacts, rays


@pytest.mark.parametrize("gl", [False, True])
def test_env_step_with_root_height_observation_vs_reference_golden(oracle, orc_char, gl):
    """`global_root_height_obs: True` (off the default config): the root height leads the character block (compute_char_obs
    ig_char_env.py:620-622), 1 313 columns -- alone and together with `global_obs`.  Fixtures: the reference's own `_post_physics_step`."""
    from helpers import default_cfg, load_clips, load_state_into, make_orc_mlib
    g = golden("env_step_root_height_obs" + ("_global" if gl else ""))
    g0 = golden("env_step")
    assert int(g["global_obs"]) == int(gl) and g["out_obs"].shape[1] == 1313
    n = g0["env_offsets"].shape[0]
    clips = load_clips([str(c) for c in g0["clips"]])
    lib = make_orc_mlib(oracle, orc_char, clips, [1.0, 1.5, 2.0, 2.5])
    ter = oracle.make_terrain(g0["hf"], g0["hf_min_point"], g0["hf_dxdy"])
    st = oracle.make_state(n, M=len(clips), obs_w=1313)
    cfg = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"], global_obs=gl, global_root_height_obs=True)
    load_state_into(st, g, "in_")
    oracle.env_post_physics_step(orc_char, lib, ter, cfg, st)
    oracle.env_update_curriculum(lib, cfg, st)
    np.testing.assert_allclose(st["obs"], g["out_obs"], atol=1e-5)
    np.testing.assert_allclose(st["reward"], g["out_reward"], atol=1e-5)
    assert np.array_equal(st["done"], g["out_done"])
    np.testing.assert_array_equal(st["obs"][:, 0], st["char_root_pos"][:, 2])      # the leading column IS the root height



@pytest.mark.parametrize("uci,eto", [(False, True), (True, False), (False, False)])
def test_env_step_without_contact_info_or_target_blocks_vs_reference_golden(oracle, orc_char, uci, eto):
    """`use_contact_info: false` (ig_parkour_env.py:72-73: no contact blocks in the observation :927-946, no contact term in the reward
    :1032-1040) and `enable_tar_obs: false` (:83: no target block, no target-contact block), alone and together (rejected with an error
    until round 4).  Fixtures: the reference's own `_post_physics_step` with those switches on the env_step.npz scene."""
    from helpers import default_cfg, load_clips, load_state_into, make_orc_mlib
    g = golden("env_step_obs_blocks_c%d_t%d" % (int(uci), int(eto)))
    g0 = golden("env_step")
    width = 136 + (630 if eto else 0) + (90 if eto and uci else 0) + (15 if uci else 0) + 441
    assert int(g["use_contact_info"]) == int(uci) and int(g["enable_tar_obs"]) == int(eto) and g["out_obs"].shape[1] == width
    n = g0["env_offsets"].shape[0]
    clips = load_clips([str(c) for c in g0["clips"]])
    lib = make_orc_mlib(oracle, orc_char, clips, [1.0, 1.5, 2.0, 2.5])
    ter = oracle.make_terrain(g0["hf"], g0["hf_min_point"], g0["hf_dxdy"])
    st = oracle.make_state(n, M=len(clips), obs_w=width)
    cfg = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"], use_contact_info=uci, enable_tar_obs=eto)
    load_state_into(st, g, "in_")
    oracle.env_post_physics_step(orc_char, lib, ter, cfg, st)
    oracle.env_update_curriculum(lib, cfg, st)
    np.testing.assert_allclose(st["obs"], g["out_obs"], atol=1e-5)
    np.testing.assert_allclose(st["reward"], g["out_reward"], atol=1e-5)
    assert np.array_equal(st["done"], g["out_done"])
    np.testing.assert_array_equal(st["fail_rates"], g["out_fail_rates"])
    assert ("out_r_contact_penalty" in g.files) == uci      # the reference has no such reward term without contact info
    # the same state under the default switches: the character block and the rays are the same columns, moved
    st2 = oracle.make_state(n, M=len(clips))
    cfg2 = default_cfg(oracle, n, g0["ray_points"], g0["env_offsets"], g0["motion_offsets"])
    load_state_into(st2, g, "in_")
    oracle.env_post_physics_step(orc_char, lib, ter, cfg2, st2)
    np.testing.assert_array_equal(st["obs"][:, :136], st2["obs"][:, :136])
    np.testing.assert_array_equal(st["obs"][:, -441:], st2["obs"][:, -441:])
    if not uci:
        assert np.abs(st["reward"] - st2["reward"]).max() > 1e-3     # the contact penalty is gone


def test_env_step_at_far_env_origins_vs_reference_golden(oracle, orc_char):
    """The env_step.npz scene with the env origins moved out by 300 m / 1 km (where the envs of a 65 536-env run sit): env-local
    root positions of hundreds of metres, one fp32 ulp = 3e-5 .. 6e-5 m.  The fixture is what the REFERENCE's own arithmetic produces there;
    the oracle restates the same operation order and lands on the same quantised values: the plain 1e-5 bar holds (measured 4.8e-7), done
    flags and fail rates exact.  The GPU test of the same fixture (tests/test_hip_parity.py) measures what the kernel's different operation
    order costs there."""
    from helpers import default_cfg, load_clips, load_state_into, make_orc_mlib
    g = golden("env_step_far")
    g0 = golden("env_step")
    n = g0["env_offsets"].shape[0]
    assert np.abs(g["in_char_root_pos"]).max() > 900.0
    clips = load_clips([str(c) for c in g0["clips"]])
    lib = make_orc_mlib(oracle, orc_char, clips, [1.0, 1.5, 2.0, 2.5])
    ter = oracle.make_terrain(g0["hf"], g0["hf_min_point"], g0["hf_dxdy"])
    st = oracle.make_state(n, M=len(clips))
    cfg = default_cfg(oracle, n, g0["ray_points"], g["env_offsets"], g0["motion_offsets"])
    load_state_into(st, g, "in_")
    oracle.env_post_physics_step(orc_char, lib, ter, cfg, st)
    oracle.env_update_curriculum(lib, cfg, st)
    err = np.abs(st["obs"] - g["out_obs"])
    assert err.max() <= 1e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))     # measured: 4.8e-7 -- the plain bar, no ulp allowance
    assert np.abs(st["reward"] - g["out_reward"]).max() <= 1e-5
    assert np.array_equal(st["done"], g["out_done"])
    np.testing.assert_array_equal(st["fail_rates"], g["out_fail_rates"])


def test_action_bounds_vs_reference_golden():
    """`_build_action_bounds_pd` (ig_char_env.py:307-347).  action_bounds.npz = the reference's own function on the joint limits
    its KinCharModel parsed from the MJCF (Isaac Gym's get_actor_dof_properties stubbed to return them): all 28 x 2 values."""
    from parc_amd.char_model import CharModel
    from parc_amd.envs import scene
    from conftest import DATA
    g = golden("action_bounds")
    cm = CharModel(os.path.join(DATA, "assets", "humanoid.xml"))
    lo, hi = cm.dof_limits()
    np.testing.assert_array_equal(np.asarray(lo, np.float32), g["dof_lower"])
    np.testing.assert_array_equal(np.asarray(hi, np.float32), g["dof_upper"])
    low, high = scene.build_action_bounds_pd(cm)
    assert low.shape == (28,) and high.shape == (28,)
    np.testing.assert_allclose(low, g["action_low"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(high, g["action_high"], rtol=0, atol=1e-7)
    assert (high > low).all()


def test_torch_full_step_vs_reference_golden(oracle, orc_char, char_golden):
    """oracle/torch_path.py::KinematicStep -- the PyTorch-CPU op sequence of the reference's WHOLE `_post_physics_step` (height rays, reference
    frame + FK, 1 312-column observation, DeepMimic reward + contact term, compute_done + motion end), which bench.py times as
    `cpu_baseline.torch_full_step` -- against the reference's own outputs on the three consecutive steps of env_step.npz."""
    import torch
    from helpers import JOINT_ERR_W, KEY_BODY_IDS, POSE_TERM_DIST, TAR_OBS_STEPS, dof_err_w_from_joint, load_clips, make_orc_mlib
    from oracle import torch_path as tp
    g = golden("env_step")
    cg = char_golden
    cm = tp.CharModel(cg["parent"], cg["local_translation"], cg["local_rotation"], cg["joint_type"], cg["joint_axis"], cg["dof_idx"], int(cg["dof_size"]))
    clips = load_clips([str(c) for c in g["clips"]])
    olib = make_orc_mlib(oracle, orc_char, clips, [1.0, 1.5, 2.0, 2.5])
    F = sum(c["root_pos"].shape[0] for c in clips)
    lib = tp.MotionLib(tp.make_tables(clips, oracle.mlib_array(olib, "frame_root_vel", (F, 3)), oracle.mlib_array(olib, "frame_root_ang_vel", (F, 3)),
                                      oracle.mlib_array(olib, "frame_dof_vel", (F, 28))))
    ws = np.array([0.5, 0.1, 0.15, 0.1, 0.15]); ws = ws / ws.sum()
    ks = tp.KinematicStep(cm, lib, tp.Terrain(g["hf"], g["hf_min_point"], g["hf_dxdy"]),
                          dict(key_body_ids=KEY_BODY_IDS, tar_obs_steps=TAR_OBS_STEPS, ray_points=g["ray_points"], env_offsets=g["env_offsets"],
                               motion_offsets=g["motion_offsets"], timestep=1.0 / 30.0, episode_length=10.0, min_obs_h=-3.0, max_obs_h=3.0,
                               reward_weights=ws, joint_err_w=JOINT_ERR_W, dof_err_w=dof_err_w_from_joint(cg, JOINT_ERR_W), contact_weights=[5.0] * 15,
                               pose_termination_dist=POSE_TERM_DIST, root_pos_termination_dist=0.6, root_rot_termination_angle=1.309))
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a))
    for s in range(3):
        p = "s%d_in_" % s
        st = {k: T(g[p + k]) for k in ("char_root_pos", "char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos", "char_dof_vel", "char_body_pos",
                                       "contact_forces", "time_offsets")}
        st["motion_ids"] = T(g[p + "motion_ids"]).long(); st["terrain_ids"] = T(g[p + "terrain_ids"]).long(); st["timestep"] = T(g[p + "timestep"]).int()
        with torch.no_grad():
            obs, rew, done = ks.step(st)
        o = "s%d_out_" % s
        err = np.abs(obs.numpy() - g[o + "obs"])
        assert err.max() <= 1e-5, (s, err.max(), np.unravel_index(err.argmax(), err.shape))
        assert np.abs(rew.numpy() - g[o + "reward"]).max() <= 1e-5
        assert np.array_equal(done.numpy(), g[o + "done"]), s
