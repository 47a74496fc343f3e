"""-m gpu: device code paths that round 1 only checked on the CPU oracle.

  * the quaternion device functions (parc_math.hpp) on the reference's edge-case vectors, through the
    test entry point ``parc_test_quat_op``;
  * the WRAP loop offset (``motion_lib.py:440-460``) through ``parc_calc_motion_frame``;
  * weighted motion sampling with the 0.01 fail-rate clamp (``motion_lib.py:56-60``, ``dm_env.py:487-490``), chi-squared;
  * cfg 2 exactly as ``dm_env_civilization.yaml`` (``terrain_build_mode: file``, one clip, 4 096 envs, kinematic step)
    against the oracle.
Tolerances: as tests/test_oracle_golden.py uses for the same vectors (the oracle is pinned to them), 1e-5 for the step.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import DATA, golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


def close(a, b, tol, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # ABSOLUTE error against tol + 2 ulp of the value (fp32: 2.4e-7 |b|) -- round 3 used a tolerance RELATIVE to max(1, |b|), i.e. 5e-4 m at 50 m;
    # the ulp term is what any fp32 evaluation order may differ by at that magnitude (the far-origin fixture measures it), tol = 0 stays exact
    err = np.abs(a - b) / (1.0 + (2.4e-7 / tol) * np.abs(b)) if tol > 0 else np.abs(a - b)
    assert np.all(np.isfinite(err)) and err.max() <= tol, f"{what}: max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"


def _qop(name, a, b=None, t=None, width=4):
    import torch
    from parc_amd import lib as L
    lib = L.load()
    dev = "cuda:0"
    ta = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    tb = None if b is None else torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(dev)
    tt = None if t is None else torch.from_numpy(np.ascontiguousarray(t, np.float32)).to(dev)
    n = ta.shape[0]
    out = torch.zeros(n, width, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.parc_test_quat_op(L.QOP[name], ta.data_ptr(), None if tb is None else tb.data_ptr(), None if tt is None else tt.data_ptr(),
                                  n, out.data_ptr(), st))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return o[:, 0] if width == 1 else o


def test_quat_device_functions_on_the_reference_edge_rows():
    g = golden("quat_ops")
    a, b, v, t, e, axis, angle = (g[k] for k in ["a", "b", "v", "t", "e", "axis", "angle"])
    T = 2e-6
    close(_qop("mul", a, b), g["quat_mul"], T, "quat_mul")
    close(_qop("mul", a, b), g["quat_multiply"], T, "quat_multiply")
    close(_qop("rotate", a, v, width=3), g["quat_rotate"], T, "quat_rotate")
    close(_qop("conj", a), g["quat_conjugate"], T, "quat_conjugate")
    close(_qop("pos", a), g["quat_pos"], T, "quat_pos")
    close(_qop("normalize3", v, width=3), g["normalize"], T, "normalize")
    aa = _qop("to_axis_angle", a)
    close(aa[:, :3], g["q2aa_axis"], T, "q2aa axis"); close(aa[:, 3], g["q2aa_angle"], T, "q2aa angle")
    close(_qop("aa_to_quat", axis, t=angle), g["aa2q"], T, "aa2q")
    close(_qop("exp_map_to_quat", e), g["exp_map_to_quat"], T, "exp_map_to_quat")
    close(_qop("to_exp_map", a, width=3), g["quat_to_exp_map"], T, "quat_to_exp_map")
    close(_qop("diff", a, b), g["quat_diff"], T, "quat_diff")
    # near-opposite/identical pairs amplify 1-ulp differences of the product through atan2 near 0/pi (same as the oracle test)
    close(_qop("diff_angle", a, b, width=1), g["quat_diff_angle"], 2e-4, "quat_diff_angle")
    close(_qop("normalize", a * np.float32(1.7)), g["quat_normalize"], T, "quat_normalize")
    close(_qop("to_tan_norm", a, width=6), g["quat_to_tan_norm"], T, "quat_to_tan_norm")
    close(_qop("slerp", a, b, t), g["slerp"], 5e-6, "slerp")
    close(_qop("heading", a, width=1), g["calc_heading"], 5e-6, "calc_heading")
    close(_qop("heading_quat_inv", a), g["calc_heading_quat_inv"], T, "calc_heading_quat_inv")
    close(_qop("rotate_2d", v, t=angle, width=2), g["rotate_2d_vec"], T, "rotate_2d_vec")


def test_reduced_range_slerp_of_the_observation_kernel():
    """k_env_post blends its 105 quaternions per env-step with slerp_rr (acos on [0, 1] and sin on [0, pi/2] by short polynomials
    instead of the general-purpose routines).  Against the reference's own slerp rows at the tolerance of the accurate
    device slerp, and against that accurate slerp on 2 M random pairs incl. the near-identical / near-opposite / tiny-angle regimes
    where the sin ratios are ill-conditioned: <= 1e-6, two decades inside the 1e-5 parity bound."""
    g = golden("quat_ops")
    close(_qop("slerp_rr", g["a"], g["b"], g["t"]), g["slerp"], 5e-6, "slerp_rr vs reference rows")
    rng = np.random.default_rng(7)
    n = 1 << 21
    a = rng.normal(size=(n, 4)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    d = rng.normal(size=(n, 4))
    scale = 10.0 ** rng.uniform(-7, 0.5, size=(n, 1))           # angle between the pair from 1e-7 rad to ~pi
    b = a + scale * d; b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[: n // 8] *= -1.0                                          # the sign flip branch
    b[n // 8: n // 8 + 1000] = a[n // 8: n // 8 + 1000]          # identical pairs: |cos| >= 1 fall-back
    t = rng.uniform(0.0, 1.0, size=n); t[:1000] = 0.0; t[1000:2000] = 1.0
    acc = _qop("slerp", a, b, t)
    rr = _qop("slerp_rr", a, b, t)
    err = np.abs(acc.astype(np.float64) - rr.astype(np.float64)).max(axis=1)
    print({"max": float(err.max()), "p99.9": float(np.quantile(err, 0.999)), "mean": float(err.mean())})
    assert np.isfinite(rr).all() and err.max() <= 1e-6, float(err.max())


def _one_clip_env(tmp_path, clip_file, n, **kw):
    from gpu_helpers import default_config
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = clip_file
    return HipParkourEnv(cfg, n, "cuda:0", False, **kw)


def test_wrap_loop_offset_on_the_device(tmp_path):
    """motion_lib_wrap.npz: civilization with loop mode WRAP, query times in [-0.5, 3] clip lengths."""
    import torch
    from parc_amd import lib as L, ms_file
    g = golden("motion_lib_wrap")
    d = ms_file.load_ms_file(os.path.join(DATA, "motion_terrains", "civilization.pkl"), load_misc=False)
    d.motion_data.loop_mode = "WRAP"
    f = str(tmp_path / "civilization.pkl")
    ms_file.save_ms_file(d, f)
    env = _one_clip_env(tmp_path, f, 8)
    assert env._scene.clips[0].loop_mode == 1
    n = len(g["q_times"])
    dev = "cuda:0"
    ids = torch.zeros(n, dtype=torch.int32, device=dev)
    tt = torch.from_numpy(g["q_times"].astype(np.float32)).to(dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    o = dict(root_pos=z(n, 3), root_rot=z(n, 4), root_vel=z(n, 3), root_ang_vel=z(n, 3), joint_rot=z(n, 14, 4), dof_vel=z(n, 28), contacts=z(n, 15))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(env._lib.parc_calc_motion_frame(env._handle, ids.data_ptr(), tt.data_ptr(), n,
                                            *[o[k].data_ptr() for k in ["root_pos", "root_rot", "root_vel", "root_ang_vel", "joint_rot", "dof_vel", "contacts"]], st))
    torch.cuda.synchronize()
    # the loop offset floor(t / len) * root_pos_delta is what distinguishes WRAP: make sure the vectors exercise it
    assert (g["q_times"] < 0).any() and (g["q_times"] > float(env._motion_lengths[0])).any()
    for k in ["root_pos", "root_rot", "joint_rot", "contacts"]:
        close(o[k].cpu().numpy(), g[k], 5e-6, k)
    for k in ["root_vel", "dof_vel"]:
        close(o[k].cpu().numpy(), g[k], 3e-5, k)  # finite-difference tables: the tolerance of test_motion_lib_vs_golden


def test_weighted_sampling_with_the_fail_rate_clamp(tmp_path):
    """p(m) ~ weight[m] * max(fail_rate[m], 0.01): weights [1, 1.5, 2, 2.5], fail rates [1, 0.5, 0.005, 0.2] (the third is
    below the clamp).  65 536 independent draws, chi-squared with 3 degrees of freedom."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 65536
    w = np.array([1.0, 1.5, 2.0, 2.5])
    fr = np.array([1.0, 0.5, 0.005, 0.2], np.float32)
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, w)
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=17, mirror_ref_state=False)
    env.set_fail_rates(fr)
    p = w * np.maximum(fr, 0.01)
    p /= p.sum()
    chi = []
    for trial in range(2):  # two independent reset calls (the Philox stream advances per call)
        env.reset()
        torch.cuda.synchronize()
        cnt = torch.bincount(env._motion_ids.long(), minlength=4).cpu().numpy().astype(np.float64)
        assert cnt.sum() == n
        chi.append(float(((cnt - n * p) ** 2 / (n * p)).sum()))
        # the clamped motion is drawn at 0.01-weight, not at its 0.005 fail rate: 4 sigma around the clamped expectation
        assert abs(cnt[2] - n * p[2]) < 4.0 * np.sqrt(n * p[2]), (cnt[2], n * p[2])
        assert cnt[2] > 1.5 * n * (w[2] * 0.005 / (w * np.array([1.0, 0.5, 0.005, 0.2])).sum())
    assert max(chi) < 21.1, chi  # chi2(3 dof) survival 1e-4
    assert np.allclose(env.get_fail_rates().numpy(), fr)  # sampling does not touch the table


@pytest.mark.parametrize("mirror", [True, False])   # False: the k_env_post<MODE, false> instantiation bench.py times
def test_cfg2_civilization_yaml_file_mode_vs_oracle(oracle, orc_char, mirror):
    """BASELINE cfg 2: 4 096 envs, the flat-terrain walk clip, terrain_build_mode: file, kinematic-only step."""
    import torch
    from gpu_helpers import to_np
    from helpers import default_cfg
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from parc_amd.util import path_loader
    n = 4096
    cfg = path_loader.load_config(os.path.join(DATA, "configs/tracker_config/dm_env_civilization.yaml"))
    assert cfg["env"]["dm"]["terrain_build_mode"] == "file" and cfg["env"]["dm"]["motion_file"].endswith("civilization.pkl")
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=5, enable_dynamics=False, mirror_ref_state=mirror)
    assert env._lib.parc_env_post_kernel(env._handle).decode() == ("k_env_post<MODE,true>" if mirror else "k_env_post<MODE,false>")
    sc = env._scene
    assert len(sc.clips) == 1 and np.array_equal(sc.grid.terrain.hf, sc.clips[0].terrain.hf) and not sc.grid.motion_offsets.any()
    env.reset()
    clips = [dict(root_pos=c.root_pos, root_rot=c.root_rot, joint_rot=c.joint_rot, contacts=c.contacts, fps=c.fps, loop_mode=c.loop_mode) for c in sc.clips]
    lib = oracle.mlib_create(orc_char, clips, [1.0])
    ocfg = default_cfg(oracle, n, sc.ray_points, sc.env_offsets, sc.grid.motion_offsets)
    ter = oracle.make_terrain(sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy)
    st = oracle.make_state(n, M=1, tracking_error=False)
    torch.manual_seed(0)
    for s in range(3):
        # SURVEY 8(d) cfg 2: char state = ref pose + N(0, 0.02^2) on root_pos and dofs, contact forces 0
        env._char_root_pos += 0.02 * torch.randn_like(env._char_root_pos)
        env._char_dof_pos += 0.02 * torch.randn_like(env._char_dof_pos)
        env._char_contact_forces.zero_()
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
        # env-local coordinates reach 64 * 10 m here: same per-row bound as test_env_step_vs_oracle_large
        pmax = np.abs(st["char_root_pos"]).max(axis=1) + 8.0
        row_tol = TOL + 2.4e-7 * pmax
        assert (err.max(axis=1) <= row_tol).all(), (s, err.max())
        near = pmax < 48.0
        assert near.sum() > 0 and err[near].max() <= TOL, err[near].max()
        rerr = np.abs(to_np(env._reward_buf) - st["reward"])
        assert (rerr <= row_tol).all() and rerr[near].max() <= TOL
        assert np.mean(to_np(env._done_buf) != st["done"]) < 1e-4
        env.reset(torch.nonzero(env._done_buf != 0).flatten())


def test_never_done_and_consumed_reset_list(tmp_path):
    """``never_done`` (ig_parkour_env.py:980): flags read NULL and reset_done() resets nobody, but the fail-rate curriculum still
    sees the episode ends.  And a reset list is consumed by reset_done(): a second call before the next step is a no-op."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml, to_np
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 2048

    def run(never_done):
        cfg = default_config()
        cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1, 1, 1, 1])
        cfg["env"]["never_done"] = never_done
        env = HipParkourEnv(cfg, n, "cuda:0", False, seed=11)
        env.reset()
        torch.manual_seed(3)
        seen = 0
        for s in range(12):
            env._char_root_pos += 0.08 * torch.randn_like(env._char_root_pos)  # drift -> pose terminations
            _, _, done, _ = env.step(None)
            d = to_np(done).copy()
            ts0 = to_np(env._timestep_buf).copy()
            env.reset_done()
            ts1 = to_np(env._timestep_buf).copy()
            if never_done:
                assert not d.any() and np.array_equal(ts0, ts1)          # nobody flagged, nobody reset
            else:
                seen += int((d != 0).sum())
                assert np.all(ts1[d != 0] == 0) and np.array_equal(ts1[d == 0], ts0[d == 0])
                ep = to_np(env._ep_num_buf).copy(); obs = to_np(env._obs_buf).copy()
                env.reset_done()                                          # the list was consumed: nothing may change
                assert np.array_equal(ep, to_np(env._ep_num_buf)) and np.array_equal(obs, to_np(env._obs_buf))
        return env.get_fail_rates().numpy(), seen
    fr_nd, _ = run(True)
    fr, seen = run(False)
    assert seen > 50
    assert (fr_nd < 1.0).any()   # the EMA ran although no flag was raised


def test_curriculum_launch_variants_are_bit_identical(tmp_path, monkeypatch):
    """Three ways to the same fail-rate table and reset list: (a) shards of up to 8 192 envs with a small library: compaction of the
    finished envs and the per-motion EMA as ONE launch (the EMA blocks read the step kernel's per-env codes themselves); (b) the two
    launches larger env counts use (k_done_scatter, then k_fail_rate_ema on the compacted list); (c) libraries of thousands of motions:
    one leader thread per motion that finished an env (k_ema_first / k_ema_leader).  All apply the reference's sequential chain
    (dm_env.py:646-660) in env order, so the tables -- and the resets drawn from them -- must be bit-identical."""
    import torch
    from gpu_helpers import default_config, write_motion_yaml
    from helpers import CLIPS4
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 8192
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = write_motion_yaml(tmp_path, CLIPS4, [1, 1, 1, 1])
    envs = []
    for switch in (None, "PARC_CURRICULUM_TWO_LAUNCHES", "PARC_EMA_LEADER"):
        if switch:
            monkeypatch.setenv(switch, "1")
        envs.append(HipParkourEnv(cfg, n, "cuda:0", False, seed=4))
        if switch:
            monkeypatch.delenv(switch, raising=False)
    for env in envs:
        env.reset()
    g = torch.Generator(device="cuda:0"); g.manual_seed(9)
    for s in range(25):
        drift = 0.06 * torch.randn(envs[0]._char_root_pos.shape, device="cuda:0", generator=g)
        for env in envs:
            env._char_root_pos += drift
            env.step(None); env.reset_done()
        fa = envs[0].get_fail_rates().numpy()
        for other in envs[1:]:
            assert np.array_equal(fa, other.get_fail_rates().numpy()), (s, fa, other.get_fail_rates().numpy())
    assert (fa < 1.0).all()
    for other in envs[1:]:
        assert torch.equal(envs[0]._motion_ids, other._motion_ids) and torch.equal(envs[0]._timestep_buf, other._timestep_buf)
        assert torch.equal(envs[0]._obs_buf, other._obs_buf)


def test_minibatch_gather_is_bit_identical_to_indexing():
    """experience_buffer.py:81-89: `sample(n)` gathers the sampled rows of every buffer with ONE HIP launch
    (parc_gather_rows) instead of one indexing kernel per buffer; byte-exact against torch indexing for float / int / bool buffers, wide and
    narrow rows, rows that are not 16-byte aligned, a ragged last block, the wrap-around of the shuffled index buffer and a partly filled
    buffer (the remainder by the sample count)."""
    import torch
    from parc_amd.learning.experience_buffer import ExperienceBuffer
    T, N = 8, 1000
    dev = "cuda:0"
    g = torch.Generator(device=dev); g.manual_seed(3)
    eb = ExperienceBuffer(T, N, dev)
    bufs = {"obs": torch.randn(T, N, 1312, device=dev, generator=g), "action": torch.randn(T, N, 28, device=dev, generator=g),
            "odd": torch.randn(T, N, 17, device=dev, generator=g),        # 68-byte rows: wide, 4-byte path
            "reward": torch.randn(T, N, device=dev, generator=g), "done": torch.randint(0, 4, (T, N), device=dev, generator=g, dtype=torch.int32),
            "mask": torch.rand(T, N, 1, device=dev, generator=g) < 0.5, "ep": torch.randint(0, 1 << 40, (T, N), device=dev, generator=g, dtype=torch.int64),
            "bytes3": torch.randint(0, 255, (T, N, 3), device=dev, generator=g, dtype=torch.uint8)}
    for k, v in bufs.items():
        eb.add_buffer(k, v)
    for filled in (T, 3):            # full buffer; 3 of 8 steps recorded (indices wrap by the sample count)
        eb.clear()
        for _ in range(filled):
            eb.inc()
        count = eb.get_sample_count()
        assert count == filled * N
        for n in (1, 255, 256, 700, 3001, 7999):    # 7999 + 3001 crosses the end of the shuffled index buffer
            head = eb._sample_buf_head
            perm = eb._sample_buf.clone()
            out = eb.sample(n)
            torch.cuda.synchronize()
            if head + n <= perm.shape[0]:
                idx = torch.remainder(perm[head:head + n], count)
                for k, v in bufs.items():
                    ref = v.view(T * N, *v.shape[2:])[idx]
                    assert out[k].dtype == v.dtype and out[k].shape == ref.shape and torch.equal(out[k], ref), (k, n, filled)
            else:   # the buffer was reshuffled inside the call: every row must still be A row of the filled part, and finite
                flat = bufs["ep"].view(-1)[:count]
                assert torch.isin(out["ep"], flat).all()
