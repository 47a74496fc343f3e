"""-m gpu: the HIP library (through its C-ABI / the HipParkourEnv shim) against the golden vectors of the real
reference and against the CPU oracle.  Tolerance: 1e-5 absolute-or-relative fp32 (BASELINE.json north_star);
integer outputs (done flags, frame indices, timesteps) exact."""
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # ABSOLUTE error against tol + 2 ulp of the value (fp32: 2.4e-7 |b|) -- round 3 used a tolerance RELATIVE to max(1, |b|), i.e. 5e-4 m at 50 m;
    # the ulp term is what any fp32 evaluation order may differ by at that magnitude (the far-origin fixture measures it), tol = 0 stays exact
    err = np.abs(a - b) / (1.0 + (2.4e-7 / tol) * np.abs(b)) if tol > 0 else np.abs(a - b)
    assert np.all(np.isfinite(err)) and err.max() <= tol, f"{what}: max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.fixture(scope="module")
def genv(tmp_path_factory):
    from gpu_helpers import golden_env
    env, g = golden_env(tmp_path_factory.mktemp("scene"), body_pos_from_fk=False, tracking=True)
    return env, g


def _ops(env):
    import ctypes as C
    import torch
    from parc_amd import lib as L

    class Ops:
        def __init__(self):
            self.lib, self.h, self.dev = env._lib, env._handle, env._device
            self.st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def t(self, a, dtype=None):
            return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev if dtype is None else self.dev, dtype=dtype)

        def dof_to_rot(self, dof):
            d = self.t(dof, torch.float32); out = torch.zeros(d.shape[0], 14, 4, device=self.dev)
            L.check(self.lib.parc_dof_to_rot(self.h, d.data_ptr(), out.data_ptr(), d.shape[0], self.st)); return out.cpu().numpy()

        def rot_to_dof(self, jr):
            j = self.t(jr, torch.float32); out = torch.zeros(j.shape[0], 28, device=self.dev)
            L.check(self.lib.parc_rot_to_dof(self.h, j.data_ptr(), out.data_ptr(), j.shape[0], self.st)); return out.cpu().numpy()

        def fk(self, rp, rr, jr):
            rp, rr, jr = self.t(rp, torch.float32), self.t(rr, torch.float32), self.t(jr, torch.float32)
            n = rp.shape[0]
            bp = torch.zeros(n, 15, 3, device=self.dev); br = torch.zeros(n, 15, 4, device=self.dev)
            L.check(self.lib.parc_forward_kinematics(self.h, rp.data_ptr(), rr.data_ptr(), jr.data_ptr(), bp.data_ptr(), br.data_ptr(), n, self.st))
            return bp.cpu().numpy(), br.cpu().numpy()

        def motion_frame(self, ids, times):
            i = self.t(ids, torch.int32); t = self.t(times, torch.float32); n = i.shape[0]
            z = lambda *s: torch.zeros(*s, device=self.dev)
            o = dict(root_pos=z(n, 3), root_rot=z(n, 4), root_vel=z(n, 3), root_ang_vel=z(n, 3), joint_rot=z(n, 14, 4), dof_vel=z(n, 28), contacts=z(n, 15))
            L.check(self.lib.parc_calc_motion_frame(self.h, i.data_ptr(), t.data_ptr(), n, *[o[k].data_ptr() for k in
                    ["root_pos", "root_rot", "root_vel", "root_ang_vel", "joint_rot", "dof_vel", "contacts"]], self.st))
            return {k: v.cpu().numpy() for k, v in o.items()}
    return Ops()


def test_library_loaded_is_in_tree(genv):
    from parc_amd import lib as L
    assert L.LIB_PATH.endswith("parc_amd/libparc_env.so")
    from parc_amd import lib as L2
    assert genv[0]._lib.parc_abi_version() == L2.ABI_VERSION == 6


def test_kin_ops_vs_golden(genv):
    env, _ = genv
    ops = _ops(env)
    g = golden("kin_ops")
    close(ops.dof_to_rot(g["dof"]), g["joint_rot"], what="dof_to_rot")
    close(ops.rot_to_dof(g["joint_rot"]), g["dof_back"], what="rot_to_dof")
    close(ops.rot_to_dof(g["joint_rot_rand"]), g["dof_rand"], what="rot_to_dof rand")
    bp, br = ops.fk(g["root_pos"], g["root_rot"], g["joint_rot"])
    close(bp, g["body_pos"], what="fk pos"); close(br, g["body_rot"], what="fk rot")
    bp, br = ops.fk(g["root_pos"], g["root_rot"], g["joint_rot_rand"])
    close(bp, g["body_pos_rand"]); close(br, g["body_rot_rand"])


def test_motion_lib_vs_golden(genv):
    """calc_motion_frame + load-time velocity tables (same 4 clips / order as motion_lib.npz)."""
    import ctypes as C
    from parc_amd import lib as L
    env, _ = genv
    ops = _ops(env)
    g = golden("motion_lib")
    F = g["frame_root_pos"].shape[0]
    rv = np.zeros((F, 3), np.float32); rav = np.zeros((F, 3), np.float32); dv = np.zeros((F, 28), np.float32)
    L.check(env._lib.parc_env_get_frame_vel_tables(env._handle, L.np_f32p(rv), L.np_f32p(rav), L.np_f32p(dv)))
    close(rv, g["frame_root_vel"], tol=1e-6, what="frame_root_vel")
    close(rav, g["frame_root_ang_vel"], tol=3e-5, what="frame_root_ang_vel")
    close(dv, g["frame_dof_vel"], tol=3e-5, what="frame_dof_vel")
    close(env._motion_lengths.cpu().numpy(), g["motion_lengths"], tol=0)
    close(env._motion_weights.cpu().numpy(), g["motion_weights"], tol=1e-7)
    o = ops.motion_frame(g["q_ids"], g["q_times"])
    for k in ["root_pos", "root_rot", "joint_rot", "contacts"]:
        close(o[k], g[k], what=k)
    for k in ["root_vel", "root_ang_vel", "dof_vel"]:
        close(o[k], g[k], tol=3e-5, what=k)


def test_known_answers_sfu(genv):
    """SURVEY §8(c) known answers (motion 0 = sfu in the golden scene)."""
    env, _ = genv
    ops = _ops(env)
    o = ops.motion_frame(np.array([0, 0]), np.array([0.10, 0.21], np.float32))
    close(o["root_pos"][0], [10.8036976, 1.8566505, 0.7566922], tol=1e-6)
    close(o["root_rot"][1], [-0.0113259, -0.0050410, 0.6899159, 0.7237490], tol=1e-6)
    bp, _ = ops.fk(o["root_pos"], o["root_rot"], o["joint_rot"])
    close(bp[0, 11], [10.9086103, 1.8863832, 0.1450104], tol=1e-6)
    close(bp[1, 8], [10.5470676, 2.4380567, 0.7927424], tol=1e-6)
    dof = ops.rot_to_dof(o["joint_rot"])
    close(dof[1, 0:6], [-0.0387498, 0.4994673, -0.0135002, 0.0099086, 0.1129964, 0.0633409], tol=1e-6)


def _check_step_outputs(env, g, p, oracle_state=None):
    from gpu_helpers import to_np
    assert np.array_equal(to_np(env._timestep_buf), g[p + "timestep"])
    close(to_np(env._time_buf), g[p + "time"], tol=0, what="time")
    for k in ["ref_root_pos", "ref_root_rot", "ref_joint_rot", "ref_body_pos", "ref_contacts", "ref_dof_pos"]:
        close(to_np(getattr(env, "_" + k)), g[p + k], what=k)
    for k in ["ref_root_vel", "ref_root_ang_vel", "ref_dof_vel"]:
        close(to_np(getattr(env, "_" + k)), g[p + k], tol=3e-5, what=k)
    ray = to_np(env._ray_hfs)
    ray_bad = np.abs(ray - g[p + "ray_hfs"]) > TOL  # a 1-ulp sin/cos difference can move a sample across a cell edge
    assert ray_bad.mean() < 2e-4, ray_bad.sum()
    obs = to_np(env._obs_buf)
    err = np.abs(obs - g[p + "obs"]); err[:, 871:][ray_bad] = 0
    assert err.max() <= TOL, (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert np.array_equal(obs[:, 871:], ray)
    close(to_np(env._reward_buf), g[p + "reward"], what="reward")
    names = ["pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "total_r"]
    rt = to_np(env._reward_terms)
    for i, nm in enumerate(names):
        close(rt[i], g[p + "r_" + nm], what=nm)
    close(to_np(env._tracking_error), g[p + "tracking_error"], what="tracking_error")
    assert np.array_equal(to_np(env._done_buf), g[p + "done"])
    close(env.get_fail_rates().numpy(), g[p + "fail_rates"], tol=0, what="fail_rates")


def test_env_step_vs_reference_golden(genv):
    """Three control steps on injected state vs the reference's own IGEnv._post_physics_step outputs."""
    from gpu_helpers import inject
    env, g = genv
    for s in range(3):
        inject(env, g, f"s{s}_in_")
        obs, rew, done, info = env.step(None)
        assert obs is env._obs_buf and rew is env._reward_buf and done is env._done_buf
        assert set(info["rewards"].keys()) == {"pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "total_r"}
        _check_step_outputs(env, g, f"s{s}_out_")


def test_fall_termination_with_contact_bodies_vs_reference_golden(tmp_path):
    """`contact_bodies: [right_foot, left_foot]` (off the default config; rejected with an error until round 3): the fall rule of
    compute_done (mgdm_dm_util.py:349-360) incl. the per-body terrain lookup (:147-152) against the reference's own
    `_post_physics_step` (env_step_fall.npz: the four combinations of contact on a non-contact body / a non-contact body below the
    termination height; pose termination off so that only the fall rule, the time limit and the motion end decide)."""
    from gpu_helpers import default_config, write_motion_yaml, inject, to_np, GOLDEN_WEIGHTS
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g, g0 = golden("env_step_fall"), golden("env_step")
    names = ["pelvis", "torso", "head", "right_upper_arm", "right_lower_arm", "right_hand", "left_upper_arm", "left_lower_arm", "left_hand",
             "right_thigh", "right_shin", "right_foot", "left_thigh", "left_shin", "left_foot"]
    outs = {}
    for fall in (True, False):
        cfg = default_config()
        cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, [str(c) for c in g0["clips"]], GOLDEN_WEIGHTS)
        cfg["env"]["hip"]["body_pos_from_fk"] = False
        cfg["env"]["pose_termination"] = bool(g["pose_termination"])
        cfg["env"]["contact_bodies"] = [names[int(b)] for b in g["contact_body_ids"]] if fall else []
        assert abs(cfg["env"]["termination_height"] - float(g["termination_height"])) < 1e-7
        env = HipParkourEnv(cfg, 64, "cuda:0", False, mirror_ref_state=fall)
        inject(env, g, "in_")
        env.step(None)
        outs[fall] = to_np(env._done_buf).copy()
        if fall:
            assert np.array_equal(outs[fall], g["out_done"])
            close(env.get_fail_rates().numpy(), g["out_fail_rates"], tol=0, what="fail_rates")
            err = np.abs(to_np(env._obs_buf) - g["out_obs"])
            ray_bad = np.abs(to_np(env._obs_buf)[:, 871:] - g["out_obs"][:, 871:]) > TOL
            err[:, 871:][ray_bad] = 0
            assert ray_bad.mean() < 2e-4 and err.max() <= TOL
    assert (g["out_done"][3::4] == 1).all()
    assert (outs[False] != outs[True]).sum() >= 8     # with contact_bodies = [] nobody falls (some of the 16 rows also reach their motion end)


def test_env_step_without_root_tracking_vs_reference_golden(tmp_path):
    """`track_root: False` (off the default config; rejected with an error until round 3): the reward in the characters' own heading
    frames (convert_to_local, mgdm_dm_util.py:247-267, :294-310) and compute_done without the root terms (:386), against the reference's
    own `_post_physics_step` (env_step_local_root.npz: sixteen characters turned by up to 2.5 rad, sixteen displaced by metres).  Runs the
    `k_env_post<STEP, *, LOCALROOT>` instantiations; the default config keeps running the ones without that code."""
    from gpu_helpers import default_config, write_motion_yaml, inject, to_np, GOLDEN_WEIGHTS
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g, g0 = golden("env_step_local_root"), golden("env_step")
    rew = {}
    for track in (False, True):
        for mirror in ((True, False) if not track else (True,)):
            cfg = default_config()
            cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, [str(c) for c in g0["clips"]], GOLDEN_WEIGHTS)
            cfg["env"]["hip"]["body_pos_from_fk"] = False
            cfg["env"]["track_root"] = track
            env = HipParkourEnv(cfg, 64, "cuda:0", False, mirror_ref_state=mirror)
            inject(env, g, "in_")
            env.step(None)
            rew[(track, mirror)] = to_np(env._reward_buf).copy()
            if not track:
                assert np.array_equal(to_np(env._done_buf), g["out_done"])
                close(to_np(env._reward_buf), g["out_reward"], what="reward")
                close(env.get_fail_rates().numpy(), g["out_fail_rates"], tol=0, what="fail_rates")
                err = np.abs(to_np(env._obs_buf) - g["out_obs"])
                ray_bad = np.abs(to_np(env._obs_buf)[:, 871:] - g["out_obs"][:, 871:]) > TOL
                err[:, 871:][ray_bad] = 0
                assert ray_bad.mean() < 2e-4 and err.max() <= TOL
    assert np.array_equal(rew[(False, True)], rew[(False, False)])             # both instantiations, bit for bit
    assert np.abs(rew[(True, True)][16:40] - rew[(False, True)][16:40]).max() > 0.05   # the switch matters on this state


def test_env_step_reward_and_done_switches_vs_reference_golden(tmp_path):
    """`track_root_h: False` and `enable_early_termination: False` (off the default config), each against the reference's own
    `_post_physics_step` (env_step_reward_done_switches.npz, see the oracle test of the same name)."""
    from gpu_helpers import default_config, write_motion_yaml, inject, to_np, GOLDEN_WEIGHTS
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g, g0 = golden("env_step_reward_done_switches"), golden("env_step")
    for tag, key in (("h0_", "track_root_h"), ("et0_", "enable_early_termination")):
        res = {}
        for on in (False, True):
            cfg = default_config()
            cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, [str(c) for c in g0["clips"]], GOLDEN_WEIGHTS)
            cfg["env"]["hip"]["body_pos_from_fk"] = False
            cfg["env"][key] = on
            env = HipParkourEnv(cfg, 64, "cuda:0", False, mirror_ref_state=False)
            inject(env, g, tag + "in_")
            env.step(None)
            res[on] = (to_np(env._reward_buf).copy(), to_np(env._done_buf).copy())
            if not on:
                assert np.array_equal(res[on][1], g[tag + "out_done"])
                close(res[on][0], g[tag + "out_reward"], what=tag + "reward")
                close(env.get_fail_rates().numpy(), g[tag + "out_fail_rates"], tol=0, what="fail_rates")
                err = np.abs(to_np(env._obs_buf) - g[tag + "out_obs"])
                ray_bad = np.abs(to_np(env._obs_buf)[:, 871:] - g[tag + "out_obs"][:, 871:]) > TOL
                err[:, 871:][ray_bad] = 0
                assert ray_bad.mean() < 2e-4 and err.max() <= TOL
        if tag == "h0_":
            assert np.abs(res[True][0][40:56] - res[False][0][40:56]).max() > 0.02
        else:
            assert (res[True][1] != 0).sum() > (res[False][1] != 0).sum() + 8


def test_env_step_with_global_observations_vs_reference_golden(tmp_path):
    """`global_obs: True` (off the default config; rejected with an error until round 3): root rotation, root velocities, root / key
    offsets of the character and of the look-ahead targets stay in the global frame (compute_char_obs ig_char_env.py:586-589, :603;
    compute_tar_obs mgdm_dm_util.py:417), against the reference's own `_post_physics_step` and reset observation
    (env_step_global_obs.npz).  Runs the `k_env_post<*, true, *, GLOBALOBS>` instantiations."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, inject, to_np, GOLDEN_WEIGHTS
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g, g0 = golden("env_step_global_obs"), golden("env_step")
    obs = {}
    for gl in (True, False):
        cfg = default_config()
        cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, [str(c) for c in g0["clips"]], GOLDEN_WEIGHTS)
        cfg["env"]["hip"]["body_pos_from_fk"] = False
        cfg["env"]["global_obs"] = gl
        env = HipParkourEnv(cfg, 64, "cuda:0", False, mirror_ref_state=False)
        assert env._lib.parc_env_post_kernel(env._handle).decode() == ("k_env_post<MODE,true>" if gl else "k_env_post<MODE,false>")
        if gl:  # the observation pass on the reference's reset state (MODE_OBS instantiation)
            inject(env, g, "reset_")
            env._compute_obs()
            torch.cuda.synchronize()
            err = np.abs(to_np(env._obs_buf) - g["reset_obs"])
            ray_bad = np.abs(to_np(env._obs_buf)[:, 871:] - g["reset_obs"][:, 871:]) > TOL
            err[:, 871:][ray_bad] = 0
            assert ray_bad.mean() < 2e-4 and err.max() <= TOL, err.max()
        inject(env, g, "in_")
        env.step(None)
        obs[gl] = to_np(env._obs_buf).copy()
        if gl:
            assert np.array_equal(to_np(env._done_buf), g["out_done"])
            close(to_np(env._reward_buf), g["out_reward"], what="reward")
            err = np.abs(obs[gl] - g["out_obs"])
            ray_bad = np.abs(obs[gl][:, 871:] - g["out_obs"][:, 871:]) > TOL
            err[:, 871:][ray_bad] = 0
            assert ray_bad.mean() < 2e-4 and err.max() <= TOL, err.max()
    d = np.abs(obs[True] - obs[False])
    assert d[:, :12].max() > 0.1 and d[:, 136:766].max() > 0.1 and d[:, 12:124].max() == 0 and d[:, 766:].max() == 0


@pytest.mark.parametrize("gl", [False, True])
def test_env_step_with_root_height_observation_vs_reference_golden(tmp_path, gl):
    """`global_root_height_obs: True` (off the default config; rejected with an error until round 3): the root height leads the character
    block (compute_char_obs ig_char_env.py:620-622) -- a 1 313-column row whose every later offset moves by one, alone and together
    with `global_obs`; against the reference's own reset observation and `_post_physics_step`."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, inject, to_np, GOLDEN_WEIGHTS
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g, g0 = golden("env_step_root_height_obs" + ("_global" if gl else "")), golden("env_step")
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, [str(c) for c in g0["clips"]], GOLDEN_WEIGHTS)
    cfg["env"]["hip"]["body_pos_from_fk"] = False
    cfg["env"]["global_root_height_obs"] = True
    cfg["env"]["global_obs"] = gl
    env = HipParkourEnv(cfg, 64, "cuda:0", False, mirror_ref_state=False)
    assert env._obs_buf.shape == (64, 1313) and env.get_obs_space().shape == (1313,)
    assert env._lib.parc_env_post_kernel(env._handle).decode() == "k_env_post<MODE,true>"

    def check(obs, ref):
        err = np.abs(obs - ref)
        ray_bad = np.abs(obs[:, 872:] - ref[:, 872:]) > TOL
        err[:, 872:][ray_bad] = 0
        assert ray_bad.mean() < 2e-4 and err.max() <= TOL, err.max()
    inject(env, g, "reset_")
    env._compute_obs()
    torch.cuda.synchronize()
    check(to_np(env._obs_buf), g["reset_obs"])
    inject(env, g, "in_")
    env.step(None)
    check(to_np(env._obs_buf), g["out_obs"])
    assert np.array_equal(to_np(env._done_buf), g["out_done"])
    close(to_np(env._reward_buf), g["out_reward"], what="reward")
    assert np.array_equal(to_np(env._obs_buf)[:, 0], to_np(env._char_root_pos)[:, 2])


@pytest.mark.parametrize("uci,eto", [(False, True), (True, False), (False, False)])
def test_env_step_without_contact_info_or_target_blocks_vs_reference_golden(tmp_path, uci, eto):
    """`use_contact_info: false` / `enable_tar_obs: false` (off the default config; rejected with an error until round 4): the contact blocks /
    the target blocks of the observation are gone and the later offsets move up (ig_parkour_env.py:927-946, mgdm_dm_util.py:482), the reward
    loses its contact term (:1032); against the reference's own reset observation and `_post_physics_step` (env_step_obs_blocks_*.npz).
    Runs the OBSVAR instantiations of k_env_post."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, inject, to_np, GOLDEN_WEIGHTS
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g, g0 = golden("env_step_obs_blocks_c%d_t%d" % (int(uci), int(eto))), golden("env_step")
    width = 136 + (630 if eto else 0) + (90 if eto and uci else 0) + (15 if uci else 0) + 441
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, [str(c) for c in g0["clips"]], GOLDEN_WEIGHTS)
    cfg["env"]["hip"]["body_pos_from_fk"] = False
    cfg["env"]["use_contact_info"] = uci
    cfg["env"]["enable_tar_obs"] = eto
    env = HipParkourEnv(cfg, 64, "cuda:0", False, mirror_ref_state=False)
    assert env._obs_buf.shape == (64, width) and env.get_obs_space().shape == (width,) and g["out_obs"].shape[1] == width
    assert env._lib.parc_env_post_kernel(env._handle).decode() == "k_env_post<MODE,true>"
    assert ("tar_obs" in env._scene.obs_shapes) == eto and ("char_contacts" in env._scene.obs_shapes) == uci
    hf0 = width - 441

    def check(obs, ref):
        err = np.abs(obs - ref)
        ray_bad = np.abs(obs[:, hf0:] - ref[:, hf0:]) > TOL
        err[:, hf0:][ray_bad] = 0
        assert ray_bad.mean() < 2e-4 and err.max() <= TOL, err.max()
    inject(env, g, "reset_")
    env._compute_obs()
    torch.cuda.synchronize()
    check(to_np(env._obs_buf), g["reset_obs"])
    inject(env, g, "in_")
    env.step(None)
    check(to_np(env._obs_buf), g["out_obs"])
    assert np.array_equal(to_np(env._done_buf), g["out_done"])
    assert np.abs(to_np(env._reward_buf) - g["out_reward"]).max() <= TOL


def test_env_step_at_far_env_origins_vs_reference_golden(tmp_path):
    """Large-N parity on the reference's own arithmetic (round-3 verdict, weak #1): the 64-env golden scene with the env origins moved out by
    300 m / 1 km, where the envs of a 65 536-env run sit -- env-local root positions up to 1 060 m, one fp32 ulp = 3e-5 .. 6e-5 m.
    env_step_far.npz is what the reference's `_post_physics_step` and reset observation produce there.  The kernel evaluates the same formulas in
    a different operation order (device sin / cos, the heading rotation applied to differences formed in another association), so a 1-ulp
    difference of a coordinate can show: the per-row bound is 1e-5 + 2 ulp(|p|max) as in test_env_step_vs_oracle_large -- here it is
    measured against the reference itself, and the test prints how many rows meet the plain 1e-5."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, inject, to_np, GOLDEN_WEIGHTS
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g, g0 = golden("env_step_far"), golden("env_step")
    for mirror in (False, True):
        cfg = default_config()
        cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, [str(c) for c in g0["clips"]], GOLDEN_WEIGHTS)
        cfg["env"]["hip"]["body_pos_from_fk"] = False
        env = HipParkourEnv(cfg, 64, "cuda:0", False, mirror_ref_state=mirror, env_offsets=g["env_offsets"])

        def check(obs, ref, pos, what):
            pmax = np.abs(pos).max(axis=1) + 8.0
            row_tol = (TOL + 2.4e-7 * pmax)[:, None]
            err = np.abs(obs - ref)
            ray_bad = err[:, 871:] > row_tol
            err[:, 871:][ray_bad] = 0
            assert ray_bad.mean() < 2e-4, (what, ray_bad.sum())
            assert (err <= row_tol).all(), (what, err.max(), np.unravel_index(err.argmax(), err.shape))
            print("far origins, %s (mirror=%s): obs max err %.2e; rows within the plain 1e-5: %d / 64" % (what, mirror, err.max(), int((err.max(axis=1) <= TOL).sum())))
            return row_tol[:, 0]
        inject(env, g, "reset_")
        env._compute_obs()
        torch.cuda.synchronize()
        check(to_np(env._obs_buf), g["reset_obs"], g["reset_char_root_pos"], "reset observation")
        inject(env, g, "in_")
        env.step(None)
        rt = check(to_np(env._obs_buf), g["out_obs"], g["in_char_root_pos"], "step")
        assert (np.abs(to_np(env._reward_buf) - g["out_reward"]) <= rt).all()
        assert (to_np(env._done_buf) != g["out_done"]).sum() <= 1     # a termination threshold compared on values that differ by an ulp
        del env


def test_env_reset_vs_reference_golden(genv):
    import torch
    from gpu_helpers import to_np
    env, g = genv
    n = env.get_num_envs()
    ep0 = to_np(env._ep_num_buf).copy()
    env.reset_with(torch.arange(n), torch.from_numpy(g["reset_motion_ids"]), torch.from_numpy(g["reset_terrain_ids"]),
                   torch.from_numpy(g["reset_time_offsets"]), torch.from_numpy(g["reset_xy_noise"]))
    for k in ["char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos", "char_dof_vel"]:
        close(to_np(getattr(env, "_" + k)), g["reset_" + k], tol=3e-5, what=k)
    close(to_np(env._char_root_pos), g["reset_char_root_pos"])
    close(to_np(env._ref_root_pos), g["reset_ref_root_pos"])
    close(to_np(env._ref_contacts), g["reset_ref_contacts"])
    assert np.array_equal(to_np(env._timestep_buf), np.zeros(n, np.int32)) and np.all(to_np(env._done_buf) == 0)
    assert np.array_equal(to_np(env._ep_num_buf), ep0 + 1)
    ray_bad = np.abs(to_np(env._ray_hfs) - g["reset_ray_hfs"]) > TOL
    assert ray_bad.mean() < 2e-4
    err = np.abs(to_np(env._obs_buf) - g["reset_obs"]); err[:, 871:][ray_bad] = 0
    assert err.max() <= TOL, err.max()
    # subset reset: untouched rows keep their observation
    before = to_np(env._obs_buf).copy()
    env.reset(torch.tensor([3, 17], device="cuda:0"))
    after = to_np(env._obs_buf)
    keep = np.ones(n, bool); keep[[3, 17]] = False
    assert np.array_equal(before[keep], after[keep])
    env.reset(torch.zeros(0, dtype=torch.long))  # empty id list is a no-op (base_agent.py:366-369)
    assert np.array_equal(after, to_np(env._obs_buf))


# mirror False = the instantiation bench.py and the learner run (k_env_post<MODE, false>: no optional ref_* / ray_hfs / tracking-error
# outputs bound); True = the one the golden-vector tests use
@pytest.mark.parametrize("mirror", [True, False])
# 1 and 100: ragged (not a multiple of the 64-env blocks of the dynamics); 13 and 1023: a short last workgroup of k_env_post (four envs
# per workgroup, the rewards on the wave (block & 3) -- here a wave that has no env, so ownership falls back to wave 0)
@pytest.mark.parametrize("n", [1, 13, 100, 1023, 4096, 16384])
def test_env_step_vs_oracle_large(tmp_path, oracle, orc_char, n, mirror):
    """Same seeded state through the HIP step and the CPU oracle at cfg-2/cfg-3 env counts (from-FK bodies)."""
    _step_vs_oracle(tmp_path, oracle, orc_char, n, mirror, {})


# the config branches off the default path (their own k_env_post instantiations), at a ragged env count: the 64-env golden fixtures pin
# their arithmetic to the reference, this pins the lane layout (short last workgroup, reward owner fall-back) against the oracle
@pytest.mark.parametrize("variant", [{"track_root": False}, {"global_obs": True}, {"track_root": False, "global_obs": True}])
def test_config_variants_vs_oracle(tmp_path, oracle, orc_char, variant):
    _step_vs_oracle(tmp_path, oracle, orc_char, 1023, False, variant)


def _step_vs_oracle(tmp_path, oracle, orc_char, n, mirror, variant):
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from helpers import CLIPS4, load_clips, make_orc_mlib, default_cfg
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    cfg = default_config()
    w = [1.0, 1.5, 2.0, 2.5]
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, w)
    cfg["env"].update(variant)
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=7, mirror_ref_state=mirror)
    general = mirror or variant.get("global_obs", False)
    assert env._lib.parc_env_post_kernel(env._handle).decode() == ("k_env_post<MODE,true>" if general else "k_env_post<MODE,false>")
    env.reset()
    rng = np.random.default_rng(0)
    steps = 3
    sc = env._scene
    clips = load_clips(CLIPS4)
    lib = make_orc_mlib(oracle, orc_char, clips, w)
    ocfg = default_cfg(oracle, n, sc.ray_points, sc.env_offsets, sc.grid.motion_offsets, **variant)
    ter = oracle.make_terrain(sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy)
    st = oracle.make_state(n, M=4, tracking_error=False)
    for s in range(steps):
        # perturb the character state like a tracking controller would leave it
        env._char_root_pos += 0.02 * torch.randn_like(env._char_root_pos)
        env._char_dof_pos += 0.05 * torch.randn_like(env._char_dof_pos)
        env._char_dof_vel += 0.2 * torch.randn_like(env._char_dof_vel)
        env._char_root_rot[:] = torch.nn.functional.normalize(env._char_root_rot + 0.02 * torch.randn_like(env._char_root_rot), dim=-1)
        f = torch.randn_like(env._char_contact_forces) * (torch.rand_like(env._char_contact_forces[..., :1]) < 0.3)
        env._char_contact_forces[:] = f
        for k_o, k_e in [("char_root_pos", "_char_root_pos"), ("char_root_rot", "_char_root_rot"), ("char_root_vel", "_char_root_vel"),
                         ("char_root_ang_vel", "_char_root_ang_vel"), ("char_dof_pos", "_char_dof_pos"), ("char_dof_vel", "_char_dof_vel"),
                         ("contact_forces", "_char_contact_forces"), ("time_offsets", "_motion_time_offsets"), ("timestep_buf", "_timestep_buf")]:
            st[k_o][...] = to_np(getattr(env, k_e))
        st["motion_ids"][...] = to_np(env._motion_ids); st["terrain_ids"][...] = to_np(env._motion_terrain_ids)
        st["fail_rates"][...] = env.get_fail_rates().numpy()
        jr = oracle.dof_to_rot(orc_char, st["char_dof_pos"])
        st["char_body_pos"][...] = oracle.forward_kinematics(orc_char, st["char_root_pos"], st["char_root_rot"], jr)[0]
        env.step(None)
        oracle.env_post_physics_step(orc_char, lib, ter, ocfg, st)
        oracle.env_update_curriculum(lib, ocfg, st)
        obs = to_np(env._obs_buf)
        ray_bad = np.abs(obs[:, 871:] - st["obs"][:, 871:]) > TOL
        assert ray_bad.mean() < 2e-4
        err = np.abs(obs - st["obs"]); err[:, 871:][ray_bad] = 0
        # With 64 envs per row the env-local coordinates reach ~300 m, where one fp32 ulp is 3e-5: the
        # reference quantises positions there exactly like we do, but a 1-ulp difference in a rotated offset
        # can flip the rounding of `root + offset`.  Bound: 1e-5 + 2 ulp(|p|max) per row.
        pmax = np.abs(st["char_root_pos"]).max(axis=1) + 8.0
        row_tol = TOL + 2.4e-7 * pmax
        assert (err.max(axis=1) <= row_tol).all(), (s, err.max(), np.unravel_index(err.argmax(), err.shape))
        near = pmax < 48.0  # rows whose coordinates stay below 48 m meet the plain 1e-5 bar
        assert near.sum() > 0 and err[near].max() <= TOL, err[near].max()
        rerr = np.abs(to_np(env._reward_buf) - st["reward"])
        assert (rerr <= row_tol).all() and rerr[near].max() <= TOL, rerr.max()
        done_h = to_np(env._done_buf)
        mism = done_h != st["done"]
        assert mism.mean() < 1e-4, mism.sum()  # threshold compares on values that differ by 1 ulp
        if not mism.any():
            close(env.get_fail_rates().numpy(), st["fail_rates"], tol=1e-6, what="fail_rates")
        done_ids = torch.nonzero(env._done_buf != 0).flatten()
        env.reset(done_ids)
    assert np.isfinite(to_np(env._obs_buf)).all()


def test_step_instantiations_bit_identical(tmp_path):
    """k_env_post<STEP, false> (what the bench times) and k_env_post<STEP, true> (what the golden tests run) on the same state at
    16 384 envs: observations, rewards, reward terms, done flags and fail rates must be bit-identical."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 16384
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1.0, 1.5, 2.0, 2.5])
    envs = [HipParkourEnv(cfg, n, "cuda:0", False, seed=21, mirror_ref_state=m) for m in (True, False)]
    assert [e._lib.parc_env_post_kernel(e._handle).decode() for e in envs] == ["k_env_post<MODE,true>", "k_env_post<MODE,false>"]
    for e in envs:
        e.reset()
    a, b = envs
    assert torch.equal(a._obs_buf, b._obs_buf) and torch.equal(a._motion_ids, b._motion_ids)   # the OBS instantiations after a reset
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    for s in range(6):
        dp = 0.03 * torch.randn(a._char_root_pos.shape, device="cuda:0", generator=gen)
        dq = 0.05 * torch.randn(a._char_dof_pos.shape, device="cuda:0", generator=gen)
        dv = 0.2 * torch.randn(a._char_dof_vel.shape, device="cuda:0", generator=gen)
        f = torch.randn(a._char_contact_forces.shape, device="cuda:0", generator=gen) * \
            (torch.rand(a._char_contact_forces[..., :1].shape, device="cuda:0", generator=gen) < 0.3)
        for e in envs:
            e._char_root_pos += dp; e._char_dof_pos += dq; e._char_dof_vel += dv; e._char_contact_forces[:] = f
            e.step(None)
        for k in ["_obs_buf", "_reward_buf", "_reward_terms", "_done_buf", "_timestep_buf", "_time_buf"]:
            assert torch.equal(getattr(a, k), getattr(b, k)), (s, k)
        assert np.array_equal(a.get_fail_rates().numpy(), b.get_fail_rates().numpy())
        assert int((a._done_buf != 0).sum()) > 0 or s < 2
        for e in envs:
            e.reset_done()
        assert torch.equal(a._obs_buf, b._obs_buf) and torch.equal(a._motion_ids, b._motion_ids)


def test_full_size_properties(tmp_path):
    """65 536 envs (BASELINE metric size): size-independent properties instead of a full oracle run."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 65536
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1, 1, 1, 1])
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=3)
    obs, info = env.reset()
    assert obs.shape == (n, 1312) and torch.isfinite(obs).all()
    # right after reset the character IS the reference pose (+xy noise): pose/vel/key rewards are exactly 1
    _, rew, done, info = env.step(None)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    # determinism: same seed -> bit-identical observation stream
    env2 = HipParkourEnv(cfg, n, "cuda:0", False, seed=3)
    obs2, _ = env2.reset()
    env2.step(None)
    assert torch.equal(obs, obs2) and torch.equal(rew, env2._reward_buf)
    # contact flags / clamp ranges
    hf = obs[:, 871:]
    assert hf.min() >= -3.0 and hf.max() <= 3.0
    cc = obs[:, 856:871]
    assert ((cc == 0) | (cc == 1)).all()
    assert set(torch.unique(done).tolist()) <= {0, 1, 3}
    # tan-norm pairs are orthonormal
    tn = obs[:, 0:6]
    assert torch.allclose((tn[:, :3] * tn[:, 3:]).sum(-1), torch.zeros(n, device=obs.device), atol=1e-5)
    assert torch.allclose(tn[:, :3].norm(dim=-1), torch.ones(n, device=obs.device), atol=1e-4)
    # sampled motions follow the weights (uniform here) within 5 sigma
    counts = torch.bincount(env._motion_ids.long(), minlength=4).float().cpu().numpy()
    assert np.all(np.abs(counts - n / 4) < 5 * np.sqrt(n * 0.25 * 0.75))
    # time offsets inside the clip
    assert (env._motion_time_offsets >= 0).all() and (env._motion_time_offsets <= env._motion_lengths[env._motion_ids.long()]).all()


def test_reset_done_device_side(tmp_path):
    """parc_env_reset_done == reset(nonzero(done)) semantics, without the host round trip."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 8192
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1, 1, 1, 1])
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=11)
    env.reset()
    total_done = 0
    for s in range(40):
        env._char_root_pos += 0.05 * torch.randn_like(env._char_root_pos)  # drift -> some envs fail
        _, _, done, _ = env.step(None)
        d = to_np(done).copy()
        ts_before = to_np(env._timestep_buf).copy(); ep_before = to_np(env._ep_num_buf).copy()
        obs_before = to_np(env._obs_buf).copy()
        env.reset_done()
        ts = to_np(env._timestep_buf); ep = to_np(env._ep_num_buf)
        was_done = d != 0
        total_done += was_done.sum()
        assert np.all(ts[was_done] == 0) and np.array_equal(ts[~was_done], ts_before[~was_done])
        assert np.array_equal(ep, ep_before + was_done)
        assert np.all(to_np(env._done_buf)[was_done] == 0)
        assert np.array_equal(to_np(env._obs_buf)[~was_done], obs_before[~was_done])
        if was_done.any():
            assert not np.array_equal(to_np(env._obs_buf)[was_done], obs_before[was_done])
    assert total_done > 100
    fr = env.get_fail_rates().numpy()
    assert np.all(fr > 0) and np.all(fr <= 1.0) and np.any(fr < 1.0)


@pytest.mark.parametrize("n", [2048, 9216])
def test_dynamics_kernel_matches_cpu_build(tmp_path, n):
    """k_dynamics_wave (HIP; 32-env blocks at 2 048 envs, 64-env blocks at 9 216) vs the host build of the same equations
    (oracle/dyn_oracle.cpp): one control step = 4 substeps, contact discovery + cached planes included.
    PhysX parity is unpinned; this checks that the GPU computes what the CPU build computes."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from helpers import CLIPS4
    from oracle.binding_dyn import DynOracle
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1, 1, 1, 1])
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=5, enable_dynamics=True)
    env.reset()
    desc = env.describe()
    assert desc["dynamics_kernel"] == "k_dynamics_wave" and desc["envs_per_block"] == ("32" if n == 2048 else "64") and desc["dev_options"] == "none"
    d = DynOracle(env._scene.cfg)
    sc = env._scene
    hf, mp, dxdy = sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(12)
    for it in range(4):
        act = (env._char_dof_pos + 0.1 * torch.randn(env._char_dof_pos.shape, device="cuda:0", generator=gen)).contiguous()
        st = dict(root_pos=to_np(env._char_root_pos).copy(), root_rot=to_np(env._char_root_rot).copy(),
                  root_vel=to_np(env._char_root_vel).copy(), root_ang_vel=to_np(env._char_root_ang_vel).copy(),
                  dof_pos=to_np(env._char_dof_pos).copy(), dof_vel=to_np(env._char_dof_vel).copy(),
                  contact_force=np.zeros((n, 15, 3), np.float32))
        env.step(act)
        d.step(hf, mp, dxdy, st, to_np(act), sc.env_offsets)
        # Evidence-based allowance (tests/diag_dyn_cpu_gpu.py, round 3, MI355X, 2 048 envs x 4 steps): NO env beyond 20 x tol, worst env
        # 7 x tol (dof_vel, step 4), 99.9 % quantile <= 0.15 tol.  An env may exceed 20 x tol only if the two builds disagree about WHICH
        # bodies are in contact (a point within rounding of a surface / cell face picked the other branch) -- checked, not assumed --
        # and even then its root stays within 1 cm; at most 0.2 % of the envs (2 x nothing, rounded up to a handful).
        cg = np.linalg.norm(to_np(env._char_contact_forces), axis=-1) > 1e-5
        cc = np.linalg.norm(st["contact_force"], axis=-1) > 1e-5
        set_differs = (cg != cc).any(1)
        for k_o, k_e, tol in [("root_pos", "_char_root_pos", 2e-4), ("root_rot", "_char_root_rot", 2e-4), ("root_vel", "_char_root_vel", 5e-3),
                              ("root_ang_vel", "_char_root_ang_vel", 2e-2), ("dof_pos", "_char_dof_pos", 1e-3), ("dof_vel", "_char_dof_vel", 5e-2)]:
            err = np.abs(to_np(getattr(env, k_e)) - st[k_o]).reshape(n, -1).max(1)
            out = err > 20 * tol
            assert np.quantile(err[~out], 0.999) <= tol, (it, k_o, np.quantile(err[~out], 0.999))
            # (9 216 envs: the host build's env-local coordinates -- ulp 3e-5 m at 400 m -- can also move a point across the MARGIN of a
            # speculative plane, which the reported forces do not show: there only the number of such envs is bounded; measured 1 of 9 216 x 4)
            assert out.mean() <= 2e-3 and (n != 2048 or set_differs[out].all()), (it, k_o, out.sum(), err[out], set_differs[out])
            # envs on whose contact set the two builds agree: a hard cap (round-3 advice).  8 x tol at 2 048 envs (measured worst 7 x tol); at
            # 9 216 envs the HOST build is the imprecise side -- it works in env-local coordinates of up to 400 m (ulp 3e-5 m), the kernel
            # in the patch frame -- measured worst 14 x tol with single envs beyond: there the quantile and the outlier count above are the check
            same = ~set_differs
            if n == 2048:
                assert err[same].max() <= 8 * tol, (it, k_o, err[same].max() / tol)
            if k_o == "root_pos":
                assert err.max() < 1e-2, (it, err.max())
        fz_g = to_np(env._char_contact_forces)[:, :, 2].sum(1); fz_c = st["contact_force"][:, :, 2].sum(1)
        assert np.quantile(np.abs(fz_g - fz_c), 0.99) < 0.02 * 500.0
        assert torch.isfinite(env._obs_buf).all() and torch.isfinite(env._reward_buf).all()


def test_dynamics_rollout_sanity(tmp_path):
    """Open-loop PD tracking of the reference pose for 2 s at 4096 envs: finite state, plausible heights/forces."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 4096
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, ["civilization"], [1.0])
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=9, enable_dynamics=True, mirror_ref_state=True)
    env.reset()
    ep_len = []
    for it in range(60):
        act = env._ref_dof_pos.clone()          # PD target = reference pose of the previous frame (open loop)
        obs, rew, done, info = env.step(act)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert torch.isfinite(env._char_root_pos).all() and torch.isfinite(env._char_dof_pos).all()
        env.reset_done()
    z = to_np(env._char_root_pos)[:, 2]
    hf = env._scene.grid.terrain.hf
    assert z.min() > hf.min() - 0.5 and z.max() < hf.max() + 3.0
    f = to_np(env._char_contact_forces)
    assert np.abs(f).max() < 1e5
    # a character that merely replays the clip's joint angles still stays up for a while: mean reward well above a fallen one
    assert to_np(env._reward_buf).mean() > 0.1


def test_ppo_training_iterations_on_hip_env(tmp_path):
    """End to end on the GPU: the PPO learner drives HipParkourEnv (full dynamics) for a few iterations through the
    reference's call sequence (reset -> [decide, step, reset done] x steps -> TD(lambda) -> minibatch updates)."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from parc_amd.learning.dm_ppo_agent import DMPPOAgent
    from parc_amd.util import path_loader
    from conftest import DATA
    import os
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1, 1, 1, 1])
    env = HipParkourEnv(cfg, 512, "cuda:0", False, seed=2, enable_dynamics=True, mirror_ref_state=False)
    acfg = path_loader.load_config(os.path.join(DATA, "configs/tracker_config/dm_agent_default.yaml"))
    acfg["steps_per_iter"] = 8
    agent = DMPPOAgent(acfg, env, "cuda:0")
    assert agent.calc_num_params() == 10638877
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    for it in range(2):
        info = agent._train_iter()
        agent._sample_count = agent._exp_buffer.get_total_samples()
        assert np.isfinite(info["loss"].item()) and np.isfinite(info["critic_loss"].item())
        assert info["env_dynamics_ms"] > 0.0 and info["env_obs_ms"] > 0.0 and 0.0 < info["roofline_frac"] < 1.0
    assert agent._obs_norm.get_count().item() == 2 * 8 * 512
    assert torch.isfinite(agent._exp_buffer.get_data("obs")).all()
    agent.save(str(tmp_path / "model.pt"))
    sd = torch.load(str(tmp_path / "model.pt"), weights_only=True)
    assert "_model._actor_layers.0.weight" in sd and "_obs_norm._mean" in sd


@pytest.mark.parametrize("n", [1, 100, 512])
def test_dynamics_kernels_agree(tmp_path, monkeypatch, n):
    """The three dynamics kernels (wave-per-limb, chain-parallel, thread-per-env) integrate the same equations: one control
    step from the same state must agree to rounding.  Guards against miscompiles of the register-heavy wave kernel."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    envs = {}
    for kern in ("wave", "coop", "thread"):
        monkeypatch.setenv("PARC_DYN_KERNEL", kern)
        cfg = default_config()
        cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, ["civilization", "sfu"], [1.0, 1.0])
        env = HipParkourEnv(cfg, n, "cuda:0", False, seed=9, enable_dynamics=True, mirror_ref_state=True)
        env.reset()
        envs[kern] = env
    names = {"wave": "k_dynamics_wave", "coop": "k_dynamics_coop", "thread": "k_dynamics"}
    for kern, env in envs.items():
        assert env._lib.parc_env_dynamics_kernel(env._handle).decode() == names[kern]
    state = ["_char_root_pos", "_char_root_rot", "_char_root_vel", "_char_root_ang_vel", "_char_dof_pos", "_char_dof_vel"]
    tol = {"_char_root_pos": 1e-4, "_char_root_rot": 1e-4, "_char_root_vel": 2e-3, "_char_root_ang_vel": 1e-2, "_char_dof_pos": 2e-4,
           "_char_dof_vel": 5e-2}
    ref = envs["wave"]
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    for it in range(4):
        act = (ref._char_dof_pos + 0.1 * torch.randn(ref._char_dof_pos.shape, device="cuda:0", generator=g)).contiguous()
        for kern in ("coop", "thread"):
            for nm in state + ["_char_contact_forces"]:
                getattr(envs[kern], nm).copy_(getattr(ref, nm))
        for env in envs.values():
            env.step(act)
        for kern in ("coop", "thread"):
            for nm in state:
                a, b = to_np(getattr(ref, nm)), to_np(getattr(envs[kern], nm))
                assert np.isfinite(a).all() and np.isfinite(b).all(), (it, kern, nm)
                assert np.abs(a - b).max() <= tol[nm], (it, kern, nm, np.abs(a - b).max())
            fa, fb = to_np(ref._char_contact_forces), to_np(envs[kern]._char_contact_forces)
            assert np.abs(fa - fb).max() <= 1.0 + 1e-3 * np.abs(fa).max(), (it, kern, np.abs(fa - fb).max())
    if n >= 512:
        # the agreement above includes the column-edge candidates of shafts / sole edges (round 3): they ARE active in this scene --
        # the same roll-out with the segments switched off (developer switch) must differ for some env
        monkeypatch.setenv("PARC_DYN_KERNEL", "wave"); monkeypatch.setenv("PARC_DYN_SEGMENTS", "none")
        cfg = default_config()
        cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, ["civilization", "sfu"], [1.0, 1.0])
        noseg = HipParkourEnv(cfg, n, "cuda:0", False, seed=9, enable_dynamics=True, mirror_ref_state=True)
        monkeypatch.delenv("PARC_DYN_SEGMENTS")
        withseg = HipParkourEnv(cfg, n, "cuda:0", False, seed=9, enable_dynamics=True, mirror_ref_state=True)
        for e in (noseg, withseg):
            e.reset()
        g2 = torch.Generator(device="cuda:0"); g2.manual_seed(3)
        differ = 0
        for it in range(12):
            act = (withseg._char_dof_pos + 0.1 * torch.randn(withseg._char_dof_pos.shape, device="cuda:0", generator=g2)).contiguous()
            for nm in state + ["_char_contact_forces"]:
                getattr(noseg, nm).copy_(getattr(withseg, nm))
            withseg.step(act); noseg.step(act)
            differ += int(((withseg._char_dof_vel - noseg._char_dof_vel).abs().amax(dim=1) > 1e-2).sum())
        assert differ > 0, "no env of the scene had a shaft / sole edge on a column edge in 12 steps"


def test_recorder_writes_motion_terrain_files(tmp_path):
    """Record mode (dm_motion_recorder.py:45-121): one env per motion, device ring buffers, files in the motion-terrain
    container.  The recorded rows must equal the state / obs history of the roll-out and the file must load back."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from parc_amd import ms_file, terrain as T
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    clips = ["sfu", "civilization"]
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, clips, [1.0, 1.0])
    out_dir = str(tmp_path / "rec")
    cfg["env"]["output_motion_dir"] = out_dir
    n = 2
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=3, enable_dynamics=True, mirror_ref_state=True)
    env.set_rand_reset(False); env.set_demo_mode(True); env.set_rand_root_pos_offset_scale(0.0)
    env._episode_length = 1000.0
    env._bypass_record_fail = True      # an open-loop PD target falls before the clip ends; the file is written anyway
    env.reset()
    env.build_agent_states_dict("_dm", record_obs=True)
    env.write_agent_states()
    hist = {e: {"root": [to_np(env._char_root_pos)[e].copy()], "rot": [to_np(env._char_root_rot)[e].copy()],
                "obs": [to_np(env._obs_buf)[e].copy()]} for e in range(n)}
    alive = [True] * n
    mids = to_np(env._motion_ids).copy()
    for it in range(400):
        if not env.is_writing_agent_states():
            break
        obs, r, done, info = env.step(env._ref_dof_pos.clone())
        d = to_np(done)
        for e in range(n):
            if alive[e]:
                hist[e]["root"].append(to_np(env._char_root_pos)[e].copy()); hist[e]["rot"].append(to_np(env._char_root_rot)[e].copy())
                hist[e]["obs"].append(to_np(obs)[e].copy())
                if d[e] == 1:
                    alive[e] = False
        env.reset_done()
    assert not env.is_writing_agent_states() and not any(alive)
    assert all(env.get_env_success_states())
    for e in range(n):
        name = clips[int(mids[e])] + "_dm.pkl"
        f = ms_file.load_ms_file(os.path.join(out_dir, name))
        k = len(hist[e]["root"])
        md = f.motion_data
        assert md.root_pos.shape == (k, 3) and md.root_rot.shape == (k, 4) and md.joint_rot.shape == (k, 14, 4) and md.body_contacts.shape == (k, 15)
        assert md.fps == 30 and md.loop_mode == "CLAMP"
        np.testing.assert_array_equal(md.root_rot, np.stack(hist[e]["rot"]))
        np.testing.assert_array_equal(f.misc_data["obs"], np.stack(hist[e]["obs"]))
        assert list(f.misc_data["obs_shapes"].keys()) == ["char_obs", "tar_obs", "tar_contacts", "char_contacts", "hf"]
        # trajectory: global xy, localised so that frame 0 is at the origin standing on hf = 0 (terrain_util.py:1617-1642)
        g = np.stack(hist[e]["root"]); g[:, 0:2] += env._scene.env_offsets[e, 0:2]
        st, loc = T.slice_terrain_around_motion(g, env._scene.grid.terrain, padding=round(1.0 // 0.4) * float(env._scene.grid.terrain.dxdy[0]))
        np.testing.assert_array_equal(md.root_pos, loc)
        assert abs(md.root_pos[0, 0]) < 1e-6 and abs(md.root_pos[0, 1]) < 1e-6
        np.testing.assert_array_equal(f.terrain_data.hf, st.hf)
        tt = T.SubTerrain.from_ms_terrain_data(f.terrain_data)
        assert tt.get_hf_val_from_points(md.root_pos[0, 0:2]) == 0.0
        assert np.isin(md.body_contacts, [0.0, 1.0]).all() and np.abs(np.linalg.norm(md.joint_rot, axis=-1) - 1.0).max() < 1e-5


def test_recorder_legacy_dict_format(tmp_path):
    """``record_format: legacy_dict``: the reference recorder's own output (ig_parkour_env.py:698-736; rows of _get_char_state :664-685):
    frames [n, 34] = localised root position | root exp map | dofs, contacts [n, 15], obs, obs_shapes, terrain."""
    import pickle
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from parc_amd.envs.hip_parkour_env import HipParkourEnv, _quat_to_exp_map_np
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, ["sfu"], [1.0])
    cfg["env"]["output_motion_dir"] = str(tmp_path / "rec")
    cfg["env"]["record_format"] = "legacy_dict"
    env = HipParkourEnv(cfg, 1, "cuda:0", False, seed=3, enable_dynamics=True, mirror_ref_state=True)
    env.set_rand_reset(False); env.set_demo_mode(True); env.set_rand_root_pos_offset_scale(0.0)
    env._episode_length = 1000.0
    env._bypass_record_fail = True
    env.reset()
    env.build_agent_states_dict("_dm", record_obs=True)
    env.write_agent_states()
    dofs, rots, obs_h, cf = [to_np(env._char_dof_pos)[0].copy()], [to_np(env._char_root_rot)[0].copy()], [to_np(env._obs_buf)[0].copy()], [to_np(env._char_contact_forces)[0].copy()]
    for it in range(200):
        if not env.is_writing_agent_states():
            break
        obs, r, done, info = env.step(env._ref_dof_pos.clone())
        dofs.append(to_np(env._char_dof_pos)[0].copy()); rots.append(to_np(env._char_root_rot)[0].copy()); obs_h.append(to_np(obs)[0].copy())
        cf.append(to_np(env._char_contact_forces)[0].copy())
        env.reset_done()
    assert not env.is_writing_agent_states()
    with open(os.path.join(str(tmp_path / "rec"), "sfu_dm.pkl"), "rb") as f:   # a file this test wrote itself
        d = pickle.load(f)
    assert set(d.keys()) == {"fps", "loop_mode", "frames", "contacts", "obs", "obs_shapes", "terrain"}
    k = len(dofs)
    assert d["fps"] == 30 and d["loop_mode"] == "CLAMP" and d["frames"].shape == (k, 34) and d["contacts"].shape == (k, 15)
    np.testing.assert_allclose(d["frames"][:, 6:], np.stack(dofs), atol=2e-6)       # rot_to_dof(dof_to_rot(dof)) on the device
    np.testing.assert_array_equal(d["frames"][:, 3:6], _quat_to_exp_map_np(np.stack(rots)))
    assert abs(d["frames"][0, 0]) < 1e-6 and abs(d["frames"][0, 1]) < 1e-6           # localised: starts at the origin
    np.testing.assert_array_equal(d["contacts"], (np.linalg.norm(np.stack(cf), axis=-1) > 1e-5).astype(np.float32))
    np.testing.assert_array_equal(d["obs"], np.stack(obs_h))
    assert list(d["obs_shapes"].keys()) == ["char_obs", "tar_obs", "tar_contacts", "char_contacts", "hf"]
    t = d["terrain"]
    assert set(t.keys()) == {"hf", "hf_maxmin", "min_point", "dxdy", "dims"} and tuple(t["dims"]) == t["hf"].shape


def test_record_mode_driver(tmp_path):
    """run_tracker --mode record == record_dm_motions(agent): an untrained policy fails every clip; with the fail filter
    bypassed each env still writes <motion>_dm.pkl, and names with a number are binned into folders of 50."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml
    from parc_amd import ms_file
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from parc_amd.learning.dm_ppo_agent import DMPPOAgent
    from parc_amd.learning.dm_motion_recorder import record_dm_motions, organize_recorded_dm_motions
    from parc_amd.util import path_loader
    from conftest import DATA
    clips = ["sfu", "civilization", "dec2024_teaser_717_1_opt_dm"]
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, clips, [1.0, 1.0, 1.0])
    cfg["env"]["output_motion_dir"] = str(tmp_path / "rec")
    env = HipParkourEnv(cfg, 3, "cuda:0", False, seed=4, enable_dynamics=True, mirror_ref_state=False)
    env._bypass_record_fail = True
    acfg = path_loader.load_config(os.path.join(DATA, "configs/tracker_config/dm_agent_default.yaml"))
    agent = DMPPOAgent(acfg, env, "cuda:0")
    ok, counts = record_dm_motions(agent, start_time_fractions=(0.1,), max_steps=600)
    assert all(ok) and counts[0] == 3
    assert env._episode_length == 1000.0 and env._demo_mode and not env._rand_reset
    got = sorted(os.listdir(str(tmp_path / "rec")))
    assert "sfu_dm.pkl" in got and "civilization_dm.pkl" in got
    assert "dec2024_teaser_700_749" in got  # dec2024_teaser_717_1_opt_dm_dm.pkl -> bin 700..749 (dm_motion_recorder.py:13-43)
    f = ms_file.load_ms_file(str(tmp_path / "rec" / "dec2024_teaser_700_749" / "dec2024_teaser_717_1_opt_dm_dm.pkl"))
    assert f.motion_data.root_pos.shape[0] >= 2 and f.misc_data["obs"].shape[1] == 1312
    # demo-mode resets start every clip at t = 0: frame 0 of the file is the clip's first pose
    src = ms_file.load_ms_file(os.path.join(DATA, "motion_terrains", "dec2024_teaser_717_1_opt_dm.pkl"))
    np.testing.assert_allclose(f.motion_data.root_rot[0], src.motion_data.root_rot[0], atol=1e-6)
    np.testing.assert_allclose(f.motion_data.joint_rot[0], src.motion_data.joint_rot[0], atol=1e-5)


def test_wide_terrain_mode_with_two_copies_per_motion(tmp_path):
    """terrain_build_mode: wide, terrains_per_motion: 2 (dm_env.py:318-445, 473-521): every env is moved onto the copy of
    its motion's terrain that it drew (`_move_to_motion_terrain`, dm_env.py:554-565)."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from parc_amd import ms_file
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from conftest import DATA
    clips = ["sfu", "civilization", "TEASER_TERRAIN"]
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, clips, [1.0, 1.0, 1.0])
    cfg["env"]["dm"]["terrain_build_mode"] = "wide"
    cfg["env"]["dm"]["terrains_per_motion"] = 2
    n = 256
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=8, enable_dynamics=False, mirror_ref_state=True)
    env.set_rand_reset(False)   # every clip starts at t = 0
    env.reset()
    mid, tid = to_np(env._motion_ids).astype(int), to_np(env._motion_terrain_ids).astype(int)
    assert set(np.unique(tid)) == {0, 1} and set(np.unique(mid)) == {0, 1, 2}
    off = env._scene.grid.motion_offsets
    assert off.shape == (3, 2, 2)
    first = np.stack([ms_file.load_ms_file(os.path.join(DATA, "motion_terrains", c + ".pkl"), load_misc=False).motion_data.root_pos[0] for c in clips])
    got = to_np(env._ref_root_pos)[:, 0:2] + env._scene.env_offsets[:, 0:2] - off[mid, tid]
    assert np.abs(got - first[mid, 0:2]).max() < 1e-4
    # the height rays see the copy's terrain: the cell under the root has the clip's own height there
    t = env._scene.grid.terrain
    g = to_np(env._ref_root_pos)[:, 0:2] + env._scene.env_offsets[:, 0:2]
    h = t.get_hf_val_from_points(g)
    src = [ms_file.load_ms_file(os.path.join(DATA, "motion_terrains", c + ".pkl"), load_misc=False) for c in clips]
    from parc_amd import terrain as T
    want = np.array([T.SubTerrain.from_ms_terrain_data(src[m].terrain_data).get_hf_val_from_points(first[m, 0:2]) for m in mid])
    np.testing.assert_array_equal(h, want)
    obs, r, d, info = env.step(None)
    assert torch.isfinite(obs).all()


def test_td_lambda_kernel_bit_exact_vs_reference_loop():
    """parc_td_lambda_return == rl_util.compute_td_lambda_return (rl_util.py:7-30), bit for bit, incl. T = 1 and done flags 1/2/3."""
    import torch
    from parc_amd.learning import rl_util
    g = torch.Generator().manual_seed(1)
    for T, N in [(1, 7), (32, 1000), (8, 4097)]:
        r = torch.rand(T, N, generator=g)
        nv = torch.randn(T, N, generator=g) * 5.0
        done = (torch.rand(T, N, generator=g) < 0.1).int() * torch.randint(1, 4, (T, N), generator=g, dtype=torch.int32)
        want = rl_util.compute_td_lambda_return_torch(r, nv, done, 0.99, 0.95)
        got = rl_util.compute_td_lambda_return(r.cuda(), nv.cuda(), done.cuda(), 0.99, 0.95).cpu()
        assert torch.equal(got, want), (T, N, (got - want).abs().max())


@pytest.mark.parametrize("dyn", [False, True])
def test_graph_step_equals_step_plus_reset_done(tmp_path, dyn):
    """parc_env_step_reset_graph replays exactly the launches of step + reset_done: state, obs, rewards, done flags, fail
    rates and the device RNG stream stay bit-identical over many steps (incl. after a setter forces a re-capture)."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 3000
    envs = []
    for _ in range(2):
        cfg = default_config()
        cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1, 2, 3, 4])
        env = HipParkourEnv(cfg, n, "cuda:0", False, seed=21, enable_dynamics=dyn, mirror_ref_state=True)
        env.reset()
        envs.append(env)
    a, b = envs
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    names = ["_obs_buf", "_reward_buf", "_done_buf", "_char_root_pos", "_char_root_rot", "_char_dof_pos", "_char_dof_vel", "_char_rigid_body_pos",
             "_motion_ids", "_motion_terrain_ids", "_motion_time_offsets", "_timestep_buf", "_ep_num_buf", "_ref_root_pos", "_char_contact_forces"]
    total_done = 0
    for it in range(60):
        if it == 30:  # a setter that is baked into the captured kernels: both paths must pick it up
            a._episode_length = 0.5; b._episode_length = 0.5
        if dyn:
            act = (a._char_dof_pos + 0.3 * torch.randn(a._char_dof_pos.shape, device="cuda:0", generator=g)).contiguous()
        else:
            act = None
            noise = 0.03 * torch.randn(a._char_root_pos.shape, device="cuda:0", generator=g)
            a._char_root_pos += noise; b._char_root_pos += noise
        a.step(act); a.reset_done()
        b.step_and_reset_done(act)
        for nm in names:
            x, y = to_np(getattr(a, nm)), to_np(getattr(b, nm))
            assert np.array_equal(x, y), (it, nm, np.abs(x.astype(np.float64) - y.astype(np.float64)).max())
    total_done = int(a._ep_num_buf.sum().item()) - n   # every reset bumps ep_num; the first one was reset()
    assert total_done > 500, total_done
    assert np.array_equal(a.get_fail_rates().numpy(), b.get_fail_rates().numpy())


def test_fused_normalize_record_bit_exact():
    """parc_normalize_record == Normalizer.normalize (normalizer.py:87-90) bit for bit, and copies the raw rows."""
    import torch
    from parc_amd.learning.normalizer import Normalizer
    g = torch.Generator().manual_seed(2)
    n, dim = 3001, 1312
    nz = Normalizer((dim,), "cuda:0", clip=10.0)
    mean = torch.randn(dim, generator=g).cuda() * 3.0
    std = (torch.rand(dim, generator=g).cuda() * 2.0 + 0.01)
    nz.set_mean_std(mean, std)
    x = (torch.randn(n, dim, generator=g) * 20.0).cuda()
    slot = torch.zeros_like(x)
    got = nz.normalize_and_record(x, slot)
    want = nz.normalize(x)
    assert torch.equal(got, want) and torch.equal(slot, x)
    assert (want.abs().max() <= nz._clip) and (want.abs() == nz._clip).any()   # the clamp is exercised
