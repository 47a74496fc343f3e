"""-m gpu: the rigid-body dynamics at the sizes the benchmark runs, and against the one PhysX artefact the
reference ships.

Isaac Gym / PhysX is a closed binary: there are no reference outputs for the integrator, PARITY IS UNPINNED.
What is checked here instead:
  * cfg 3 (16 384 envs, 1 024-entry library on the blocky grid) and the headline 65 536 envs with the dynamics ON:
    finite state, bounded velocities / forces, resting contact force = weight, kinetic energy decays at rest;
  * k_dynamics_wave against k_dynamics_coop on a 4 096-env slice of the cfg-3 scene (same state, same actions);
  * an open-loop replay of ``dec2024_teaser_717_1_opt_dm.pkl`` — a trajectory PhysX produced under this PD model, recorded
    by ``ig_parkour_env.py:759-796`` — with PD targets = its own next-frame dofs: steps until pose termination, agreement
    of the simulated contact flags with the recorded ``body_contacts``, root-height error.  The thresholds are what this
    simulator achieves today; they pin it against regressions, they are not PhysX parity.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg3(tmp_path, motions):
    from gpu_helpers import default_config
    from conftest import DATA
    from parc_amd.util import synth_dataset
    cfg = default_config()
    if motions:
        cfg["env"]["dm"]["motion_file"] = synth_dataset.write_spec(str(tmp_path / "motions.yaml"),
                                                                   os.path.join(DATA, "motion_terrains", "motions_bundled.yaml"), motions, yaw=False)
    return cfg


def _bench_actions(env, gen):
    import torch
    lo, hi = env._action_bound_low, env._action_bound_high
    mean, std = 0.5 * (hi + lo), 0.5 * (hi - lo)
    return mean + 0.05 * std * torch.randn(env._char_dof_pos.shape, device=env._device, generator=gen)


def _finite_state(env):
    import torch
    for nm in ("_char_root_pos", "_char_root_rot", "_char_root_vel", "_char_root_ang_vel", "_char_dof_pos", "_char_dof_vel", "_char_contact_forces",
               "_obs_buf", "_reward_buf"):
        assert torch.isfinite(getattr(env, nm)).all(), nm


@pytest.mark.parametrize("n,motions", [(16384, 1024), (65536, 0)])
def test_dynamics_at_bench_sizes(tmp_path, n, motions):
    """The benchmark's own scenario (untrained-policy actions, step + reset_done) and a settle-to-rest run, dynamics ON."""
    import torch
    from gpu_helpers import to_np
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    env = HipParkourEnv(_cfg3(tmp_path, motions), n, "cuda:0", False, seed=21, enable_dynamics=True, mirror_ref_state=False)
    assert env._lib.parc_env_dynamics_kernel(env._handle).decode() == "k_dynamics_wave"
    assert len(env._scene.clips) == (motions or 5)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(4)
    env.reset()
    hf = env._scene.grid.terrain.hf
    for it in range(30):  # (1) the bench scenario
        env.step(_bench_actions(env, gen))
        if it % 10 == 9:
            _finite_state(env)
            assert float(env._char_root_vel.norm(dim=-1).max()) < 60.0
            assert float(env._char_dof_vel.abs().max()) <= 100.0 + 1e-3          # max_angular_velocity clamp
            assert float(env._char_contact_forces.abs().max()) < 2.0e5
        env.reset_done()
    z = to_np(env._char_root_pos)[:, 2]
    assert z.min() > hf.min() - 1.0 and z.max() < hf.max() + 4.0
    rq = to_np(env._char_root_rot)
    assert np.abs(np.linalg.norm(rq, axis=1) - 1.0).max() < 1e-4
    # (2) settle: hold the reset pose with zero velocity for 8 s, no resets.  Whatever a character ends up doing (standing, kneeling,
    # lying at the foot of a wall), at rest its contact forces carry its weight and its kinetic energy is gone.  Coming to rest takes
    # long for a held-pose mannequin on this terrain: it balances, topples (the median env needs 2 s), and a few percent then tumble down
    # the ledges of pits up to 13.6 m deep.  tools/dyn_settle_diag.py classifies the envs outside +-25 %: at 4 s they
    # are still moving (falling, tumbling with one or two contacts), at 8 s 0.8 % are left (round 3: 0.935 / 0.992; round 2: 0.930 / 0.982).
    env.reset()
    env._char_root_vel.zero_(); env._char_root_ang_vel.zero_(); env._char_dof_vel.zero_()
    hold = env._char_dof_pos.clone()
    speed = []
    mg = 9.81 * 50.05  # humanoid.xml: 50.05 kg
    within = {}
    for it in range(240):
        env.step(hold)
        speed.append(float(env._char_root_vel.norm(dim=-1).median()))
        if it in (119, 239):
            within[it] = float(np.mean(np.abs(to_np(env._char_contact_forces)[:, :, 2].sum(1) / mg - 1.0) < 0.25))
    _finite_state(env)
    fz = to_np(env._char_contact_forces)[:, :, 2].sum(1) / mg
    assert 0.97 < np.median(fz) < 1.03, np.median(fz)
    assert within[119] > 0.90 and within[239] > 0.98, within
    fxy = np.abs(to_np(env._char_contact_forces)[:, :, :2].sum(1)) / mg
    assert np.median(fxy) < 0.25, np.median(fxy)                                  # friction cone: |F_t| <= mu F_n with mu = 1
    assert speed[-1] < 0.05 and speed[-1] < 0.1 * max(speed), (speed[-1], max(speed))  # kinetic energy is gone
    sp = to_np(env._char_root_vel.norm(dim=-1))
    assert np.quantile(sp, 0.99) < 1.0 and sp.max() < 30.0, (np.quantile(sp, 0.99), sp.max())   # after 8 s 99 % are at rest (measured q99 0.11 m/s); nobody is launched
    assert np.mean(np.abs(to_np(env._char_contact_forces)).reshape(n, -1).max(1) > 20 * mg) < 5e-3
    assert env._lib.parc_env_dynamics_timeouts(env._handle) == 0   # no flag wait between the waves of a block ever hit its bound
    assert env.dynamics_manifold_drops() == 0                      # every contact plane found a slot (LDS share + overflow area of its wave)


def test_wave_kernel_vs_coop_on_a_slice_of_the_cfg3_scene(tmp_path, monkeypatch):
    """Permanent guard for the register-heavy wave kernel (its -fno-slp-vectorize history): at the
    benchmark's scale, on the cfg-3 scene, the first 4 096 envs must evolve like the independent chain-parallel kernel."""
    import torch
    from gpu_helpers import to_np
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n, ns = 16384, 4096
    cfg = _cfg3(tmp_path, 1024)
    big = HipParkourEnv(cfg, n, "cuda:0", False, seed=33, enable_dynamics=True, mirror_ref_state=False)
    monkeypatch.setenv("PARC_DYN_KERNEL", "coop")
    small = HipParkourEnv(cfg, ns, "cuda:0", False, seed=33, enable_dynamics=True, mirror_ref_state=False, env_id_base=0, total_envs=n)
    monkeypatch.delenv("PARC_DYN_KERNEL")
    assert big._lib.parc_env_dynamics_kernel(big._handle).decode() == "k_dynamics_wave"
    assert small._lib.parc_env_dynamics_kernel(small._handle).decode() == "k_dynamics_coop"
    assert np.array_equal(big._scene.env_offsets[:ns], small._scene.env_offsets)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(8)
    big.reset()
    hand_contacts = 0
    for it in range(12):  # let the population spread out (falls, wall contacts, resets)
        big.step(_bench_actions(big, gen)); big.reset_done()
    state = ["_char_root_pos", "_char_root_rot", "_char_root_vel", "_char_root_ang_vel", "_char_dof_pos", "_char_dof_vel"]
    sync = state + ["_char_contact_forces", "_motion_ids", "_motion_terrain_ids", "_motion_time_offsets", "_timestep_buf"]
    tol = {"_char_root_pos": 1e-4, "_char_root_rot": 1e-4, "_char_root_vel": 2e-3, "_char_root_ang_vel": 1e-2, "_char_dof_pos": 2e-4,
           "_char_dof_vel": 5e-2}
    for it in range(3):
        for nm in sync:
            getattr(small, nm).copy_(getattr(big, nm)[:ns])
        act = _bench_actions(big, gen)
        big.step(act); small.step(act[:ns].contiguous())
        for nm in state:
            a, b = to_np(getattr(big, nm))[:ns], to_np(getattr(small, nm))
            assert np.isfinite(a).all() and np.isfinite(b).all(), (it, nm)
            err = np.abs(a - b).reshape(ns, -1).max(1)
            if nm == "_char_root_pos":  # the coop kernel integrates in env-local coordinates (hundreds of metres here, ulp 3e-5..6e-5)
                err = np.maximum(err - 4.0 * 1.2e-7 * np.abs(a).max(1), 0.0)  # and rounds every substep; the wave kernel does not
            # a contact that exists in one kernel and not in the other (a point within rounding of a surface or of a cell face)
            # moves single envs; everything else agrees to rounding
            q = (np.quantile(err, 0.99), np.quantile(err, 0.999), err.max())
            assert q[0] <= tol[nm] and np.mean(err > 10 * tol[nm]) < 5e-3, (it, nm, q, np.mean(err > 10 * tol[nm]))
        fa, fb = to_np(big._char_contact_forces)[:ns], to_np(small._char_contact_forces)
        ferr = np.abs(fa - fb).reshape(ns, -1).max(1)
        assert np.quantile(ferr, 0.99) <= 1.0 + 1e-3 * np.abs(fa).max(), (it, np.quantile(ferr, 0.99), ferr.max())
        # the hands (fixed joints) are merged into the lower arms inside k_dynamics_wave (build_wave_tables) and still report their own
        # contact force: their flags must agree with the kernel that keeps them as bodies of their own
        hands = [5, 8]
        ha, hb = np.linalg.norm(fa[:, hands], axis=-1) > 1e-5, np.linalg.norm(fb[:, hands], axis=-1) > 1e-5
        assert (ha == hb).mean() > 0.998, (it, (ha == hb).mean())
        hand_contacts += int(hb.sum())
        assert np.array_equal(to_np(big._done_buf)[:ns] != 0, to_np(small._done_buf) != 0) or \
            np.mean((to_np(big._done_buf)[:ns] != 0) != (to_np(small._done_buf) != 0)) < 1e-3
    assert hand_contacts > 50, hand_contacts     # the scenario does put hands on the ground


def physx_replay_metrics(n=32, clip="dec2024_teaser_717_1_opt_dm"):
    """Open-loop replay (see the module docstring).  Env i starts at fraction i / n * 0.8 of the clip."""
    import ctypes as C
    import torch
    from gpu_helpers import default_config, to_np
    from conftest import DATA
    from parc_amd import lib as L
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = os.path.join(DATA, "motion_terrains", clip + ".pkl")
    cfg["env"]["dm"]["terrain_build_mode"] = "file"
    cfg["env"]["rand_reset"] = False
    cfg["env"]["rand_root_pos_offset_scale"] = 0.0
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1, enable_dynamics=True, mirror_ref_state=True)
    frac = torch.linspace(0.0, 0.8, n, device="cuda:0")
    env.set_reset_motion_start_time_fraction(frac)
    env.reset()
    dt = 1.0 / 30.0
    length = float(env._motion_lengths[0])
    nsteps = int(round(length / dt))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ids = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    z = lambda *s: torch.zeros(*s, device="cuda:0")
    o = dict(root_pos=z(n, 3), root_rot=z(n, 4), root_vel=z(n, 3), root_ang_vel=z(n, 3), joint_rot=z(n, 14, 4), dof_vel=z(n, 28), contacts=z(n, 15))
    target = z(n, 28)
    first_fail = np.full(n, -1)
    end_step = np.full(n, -1)
    hist = []
    for k in range(nsteps):
        t_next = (env._timestep_buf.float() + 1.0) * dt + env._motion_time_offsets
        L.check(env._lib.parc_calc_motion_frame(env._handle, ids.data_ptr(), t_next.contiguous().data_ptr(), n,
                                                *[o[q].data_ptr() for q in ["root_pos", "root_rot", "root_vel", "root_ang_vel", "joint_rot", "dof_vel", "contacts"]], st))
        L.check(env._lib.parc_rot_to_dof(env._handle, o["joint_rot"].data_ptr(), target.data_ptr(), n, st))
        _, rew, done, _ = env.step(target)
        mt = to_np((env._timestep_buf.float()) * dt + env._motion_time_offsets)
        ended = mt >= length - 1e-4
        d = to_np(done)
        for i in range(n):
            if ended[i] and end_step[i] < 0:
                end_step[i] = k
            if d[i] == 1 and not ended[i] and first_fail[i] < 0:
                first_fail[i] = k
        sim_c = to_np(env._char_contact_forces.norm(dim=-1) > 1e-5)
        ref_c = to_np(env._ref_contacts) > 0.5
        hist.append(dict(zerr=np.abs(to_np(env._char_root_pos)[:, 2] - to_np(env._ref_root_pos)[:, 2]),
                         xyerr=np.linalg.norm(to_np(env._char_root_pos)[:, :2] - to_np(env._ref_root_pos)[:, :2], axis=1),
                         agree=(sim_c == ref_c), ref_c=ref_c, sim_c=sim_c, rew=to_np(rew).copy(), ended=ended.copy()))
        assert torch.isfinite(env._char_root_pos).all()
    # per env: the stretch it tracked = steps before the first pose termination (or before the clip ended)
    tracked = np.where(first_fail >= 0, first_fail, np.where(end_step >= 0, end_step, nsteps))
    ok = np.zeros((nsteps, n), bool)
    for i in range(n):
        ok[: tracked[i], i] = True
    zerr = np.stack([h["zerr"] for h in hist])[ok]
    agree = np.stack([h["agree"] for h in hist])[ok]          # [samples, 15]
    ref_c = np.stack([h["ref_c"] for h in hist])[ok]
    sim_c = np.stack([h["sim_c"] for h in hist])[ok]
    feet = [11, 14]
    return dict(steps_total=nsteps, tracked=tracked, survived_to_end=(first_fail < 0), zerr_mean=float(zerr.mean()), zerr_q90=float(np.quantile(zerr, 0.9)),
                contact_agreement_all=float(agree.mean()), contact_agreement_feet=float(agree[:, feet].mean()),
                foot_contact_rate_ref=float(ref_c[:, feet].mean()), foot_contact_rate_sim=float(sim_c[:, feet].mean()),
                reward_mean_tracked=float(np.stack([h["rew"] for h in hist])[ok].mean()))


def test_physx_recorded_trajectory_open_loop_replay():
    m = physx_replay_metrics()
    print({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in m.items()})
    # thresholds = what the simulator does today with margin (DESIGN.md section 2 quotes the measured values)
    # measured (round 2, MI355X): median 18 steps, z error 0.098 m mean / 0.30 m q90, flags 0.943 all bodies / 0.730 feet,
    # foot contact rate 0.49 simulated vs 0.61 recorded, mean reward 0.955 while tracking
    assert np.median(m["tracked"]) >= 12            # control steps of open-loop tracking before pose termination
    assert m["zerr_mean"] < 0.15                    # m, root height vs the PhysX trajectory while tracking
    assert m["contact_agreement_all"] > 0.90        # simulated contact flags vs the recorded body_contacts, all bodies
    assert m["contact_agreement_feet"] > 0.65
    assert abs(m["foot_contact_rate_sim"] - m["foot_contact_rate_ref"]) < 0.25
    assert m["reward_mean_tracked"] > 0.8


def test_cfg5_shard_synthetic_full_dataset_shape(tmp_path):
    """BASELINE cfg 5 on one GPU's shard: 16 384 envs (= 131 072 / 8) on the M = 16 384 pseudo-clip library of SURVEY 8(d)
    (bundled clip i mod 5, yaw 2 pi i / M, weight = clip length), per-env motion-terrain offsets + curriculum resets, dynamics on.
    Size-independent properties: the library is what the generator promises, resets land the character on ITS tile of the
    16 384-tile grid standing on the (rotated) ground, sampling follows the weights, the step stays finite and deterministic."""
    import torch
    from gpu_helpers import default_config, to_np
    from conftest import DATA
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from parc_amd.util import synth_dataset
    n, M = 16384, 16384
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = synth_dataset.write_spec(str(tmp_path / "motions.yaml"), os.path.join(DATA, "motion_terrains", "motions_bundled.yaml"), M, yaw=True)
    mk = lambda: HipParkourEnv(cfg, n, "cuda:0", False, seed=77, enable_dynamics=True, mirror_ref_state=True, env_id_base=0, total_envs=131072)
    env = mk()
    sc = env._scene
    assert len(sc.clips) == M and sc.grid.motion_offsets.shape == (M, 1, 2)
    lengths = np.array([(c.num_frames - 1) / c.fps for c in sc.clips])
    assert np.allclose([c.weight for c in sc.clips], lengths)                       # weight = clip length
    assert np.allclose(to_np(env._motion_lengths), lengths, atol=1e-5)
    yaw0 = 2.0 * np.arctan2(sc.clips[0].root_rot[0, 2], sc.clips[0].root_rot[0, 3])  # clip 5 k = clip 0 turned by 2 pi 5k / M
    k = 1000
    yawk = 2.0 * np.arctan2(sc.clips[5 * k].root_rot[0, 2], sc.clips[5 * k].root_rot[0, 3])
    assert abs(((yawk - yaw0) - 2.0 * np.pi * 5 * k / M + np.pi) % (2.0 * np.pi) - np.pi) < 2e-2
    tables_mb = (sum(c.num_frames for c in sc.clips) * 512 + sc.grid.terrain.hf.nbytes) / 1e6
    assert tables_mb > 512.0, tables_mb   # frame records + grid exceed the 256 MB Infinity Cache: the table-miss row of SURVEY 8(d)
    obs, _ = env.reset()
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all()
    # every env sits on the tile of ITS motion: global xy inside the tile's footprint
    mid = to_np(env._motion_ids).astype(np.int64)
    gxy = to_np(env._char_root_pos)[:, :2] + sc.env_offsets[:, :2]
    ter = np.array([[c.terrain.min_point[0], c.terrain.min_point[1], c.terrain.hf.shape[0] * c.terrain.dx, c.terrain.hf.shape[1] * c.terrain.dx] for c in sc.clips])
    lo = sc.grid.motion_offsets[mid, 0] + ter[mid, :2]
    assert np.all(gxy >= lo - 0.5) and np.all(gxy <= lo + ter[mid, 2:] + 0.5)
    # and stands on the rotated ground: the height sample under the root (ray point (0, 0) is index 2 of every 63-point ray)
    under = to_np(env._ray_hfs)[:, 3 * 63 + 2]
    # (the bundled clips cross gaps: in 22-30 % of their frames the root is over a drop deeper than the 3 m clamp)
    assert -1.1 < np.median(under) < -0.6 and np.quantile(under, 0.95) < -0.3 and np.mean(under <= -2.99) < 0.36, \
        (np.median(under), np.quantile(under, 0.95), np.mean(under <= -2.99))
    # sampling ~ weight (fail rates are 1): the five base clips' shares
    share = np.array([lengths[mid % 5 == b].size for b in range(5)], np.float64) / n
    expect = np.array([lengths[b::5].sum() for b in range(5)]) / lengths.sum()
    assert np.abs(share - expect).max() < 5.0 * np.sqrt(0.25 / n) + 1e-3, (share, expect)
    assert len(np.unique(mid)) > 0.55 * n                                          # 16 384 draws from 16 384 entries: ~63 % distinct
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    acts = [_bench_actions(env, gen) for _ in range(6)]
    for a in acts:
        env.step(a); env.reset_done()
    _finite_state(env)
    env2 = mk()
    env2.reset()
    for a in acts:
        env2.step(a); env2.reset_done()
    assert torch.equal(env._obs_buf, env2._obs_buf) and torch.equal(env._motion_ids, env2._motion_ids)  # same seed: bit-identical
    fr = env.get_fail_rates().numpy()
    assert fr.shape == (M,) and fr.min() > 0.0 and fr.max() <= 1.0 and (fr < 1.0).any()


def _far_vs_near_drift(monkeypatch, residual):
    """65 536 identical characters (same clip, same frame, no noise) pushed sideways at 3 mm/s under zero gravity: every env
    must travel the same distance.  Returns the travelled x distance after 2 s for the env at the origin of the env grid and
    for the far corner (env-local coordinates ~1 km, ulp 6e-5 m), plus the expected distance."""
    import torch
    from gpu_helpers import default_config, to_np
    from conftest import DATA
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    if not residual:
        monkeypatch.setenv("PARC_DYN_NO_RESIDUAL", "1")
    n = 65536
    cfg = default_config()
    cfg["env"]["dm"]["motion_file"] = os.path.join(DATA, "motion_terrains", "civilization.pkl")
    cfg["env"]["gravity_z"] = 0.0
    env = HipParkourEnv(cfg, n, "cuda:0", False, seed=1, enable_dynamics=True, mirror_ref_state=False)
    monkeypatch.delenv("PARC_DYN_NO_RESIDUAL", raising=False)
    ids = torch.arange(n, device="cuda:0")
    zi = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    env.reset_with(ids, zi, zi, torch.full((n,), 1.0, device="cuda:0"), torch.zeros(n, 2, device="cuda:0"))
    env._char_root_pos[:, 2] += 1.0   # lift everybody clear of the ground: a free-floating body keeps its momentum
    env._char_root_vel.zero_(); env._char_root_ang_vel.zero_(); env._char_dof_vel.zero_()
    env._char_root_vel[:, 0] = 3.0e-3
    hold = env._char_dof_pos.clone()
    off = env._scene.env_offsets
    far = int(np.argmax(np.abs(off[:, 0]) + np.abs(off[:, 1])))
    x0 = to_np(env._char_root_pos).astype(np.float64)[:, 0].copy()
    for it in range(60):
        env.step(hold)
    x1 = to_np(env._char_root_pos).astype(np.float64)[:, 0]
    assert abs(float(env._char_root_pos[far, 0])) > 500.0  # the far corner really is far in env-local coordinates
    return (x1 - x0)[0], (x1 - x0)[far], 3.0e-3 * 2.0


def test_far_envs_integrate_as_precisely_as_near_ones(monkeypatch):
    """Env-local fp32 coordinates reach ~1 km at 65 536 envs.  The dynamics integrates in the frame of the root's cell and keeps
    what the fp32 write-back rounds away, so a 3 mm/s drift is not lost 1 km from the origin."""
    near, far, expect = _far_vs_near_drift(monkeypatch, residual=True)
    before = _far_vs_near_drift(monkeypatch, residual=False)
    print({"with_residual": (near, far), "without": before[:2], "expected": expect})
    assert abs(near - expect) < 0.05 * expect
    assert abs(far - expect) < 0.05 * expect + 6.1e-5           # one rounding of the buffer value itself
    assert abs(before[1] - expect) > abs(far - expect)         # the per-step rounding this removes (DESIGN.md quotes the numbers)


def test_step_observation_equals_recomputed_observation_bit_for_bit():
    """After a step the observation kernel reads the prep records (heading terms, dof -> quat of the character) that
    k_dynamics_wave wrote with the state; parc_env_compute_obs forms them with k_env_prep from the stored state.  Same device
    functions on the same fp32 inputs: the two observation rows must be identical bits, for every env."""
    import torch
    from gpu_helpers import default_config
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 4096
    env = HipParkourEnv(default_config(), n, "cuda:0", False, seed=5, enable_dynamics=True, mirror_ref_state=False)
    env.reset()
    lo, hi = env._action_bound_low, env._action_bound_high
    g = torch.Generator(device="cuda:0"); g.manual_seed(11)
    for it in range(6):
        act = (0.5 * (hi + lo) + 0.3 * 0.5 * (hi - lo) * torch.randn(n, env._char_dof_pos.shape[1], device="cuda:0", generator=g)).contiguous()
        obs, _, _, _ = env.step(act)
        stepped = obs.clone()
        again = env._compute_obs().clone()     # k_env_prep + the observation kernel on the state the step left behind
        torch.cuda.synchronize()
        assert torch.equal(stepped.view(torch.int32), again.view(torch.int32)), f"step {it}: {(stepped != again).sum().item()} values differ"
        env.reset_done()


def test_bench_launches_its_own_ranks_on_the_gpu():
    """`python bench.py --gpus 2` end to end on hardware: the launcher starts two fresh ranks (both on this box's one GPU:
    PARC_BENCH_SHARE_GPU, gloo for the barrier / max), each owns 4 096 of 8 192 envs (strong scaling) with the env origins of
    its global index range, rank 0 prints exactly one JSON line with the contract's fields."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, PARC_BENCH_SHARE_GPU="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--envs", "8192", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 20 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["config"]["total_envs"] == 8192 and d["config"]["envs_per_gpu"] == 4096 and "workload" in d["config"]
    assert abs(d["value"] - 8192 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"] == "k_dynamics_wave" and d["roofline"]["kernel_ms"] > 0 and 0 < d["roofline"]["frac"] < 1
    assert 1e6 < d["value"] < 1e9


def test_bench_ppo_leg_runs_over_rccl_on_one_gpu():
    """cfg 4's collective leg: `bench.py --ppo 1` at one rank forms a 1-rank `nccl` (= RCCL) process group and issues the 40 x 42.56 MB
    gradient all-reduces every 32 steps, so the RCCL path is loaded, executed and timed on this box.
    65 steps = two PPO iterations' worth of collectives."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PARC_BENCH_SHARE_GPU", "PARC_BENCH_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--envs", "8192", "--steps", "65", "--warmup", "32",
                        "--ppo", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")][-1])
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["allreduce_bytes"] == 42555508 and c["allreduces_per_iter"] == 40
    assert c["iters_timed"] == 2 and c["ms_per_iter_leg"] > 0.0
    assert "all-reduces" in d["config"]["workload"] and d["dynamics_timeouts"] == 0 and d["n_gpus"] == 1


_BREAK_BODY = r"""
import sys, torch
sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + "/tests")
from gpu_helpers import default_config
from parc_amd import lib as L
from parc_amd.envs.hip_parkour_env import HipParkourEnv
assert L.LIB_PATH.endswith("libparc_env_breakflag.so")
env = HipParkourEnv(default_config(), 256, "cuda:0", False, seed=5, enable_dynamics=True, mirror_ref_state=False)
env.reset()
assert env.dynamics_timeouts() == 0 and torch.isfinite(env._char_root_pos).all()
obs, rew, done, info = env.step(env._char_dof_pos.clone())
torch.cuda.synchronize()
n_to = env.dynamics_timeouts()
assert n_to >= 4, n_to                                     # every block: wave 0 waits for the flag that is never published (once a block has timed out its later waits give up at once)
assert env._health is not None and int(env._health[0]) == n_to   # the host-mapped copy, refreshed by the launch that closed the step
try:                                                       # ... which the NEXT step reads without a synchronisation: the run stops there
    env.step(env._char_dof_pos.clone())
except L.ParcError as ex:
    assert "timed out" in str(ex)
else:
    raise AssertionError("step() must raise once the host-mapped health counter is non-zero")
assert torch.isnan(env._char_root_pos).all()               # every block saw a timeout: every env is poisoned ...
assert torch.isnan(obs).any(dim=1).all() and torch.isnan(rew).all()   # ... and so is what the learner would read
try:
    env.get_extra_log_info()
except L.ParcError as ex:
    assert "timed out" in str(ex)
else:
    raise AssertionError("get_extra_log_info() must raise when the health counter is non-zero")
print("BREAKFLAG_OK", n_to)
"""


def test_flag_timeout_is_counted_and_poisons_the_state(tmp_path):
    """A hand-off of k_dynamics_wave that never arrives (test build -DPARC_TEST_BREAK_FLAG: the producer of one LDS flag does not
    publish; short wait bound) must not go unnoticed: the device counter counts it, the block writes NaN root positions, the
    observation / reward of those envs are NaN, and HipParkourEnv.get_extra_log_info() raises (run_tracker.py then exits non-zero).
    One run, in a child process (the library path is fixed at import)."""
    import subprocess
    import sys
    from conftest import REPO
    lib = os.path.join(REPO, "parc_amd", "libparc_env_breakflag.so")
    assert os.path.exists(lib), "build it with __graft_entry__.build()"
    script = tmp_path / "breakflag.py"
    script.write_text(_BREAK_BODY.format(repo=REPO))
    p = subprocess.run([sys.executable, str(script)], env=dict(os.environ, PARC_ENV_LIB=lib, PARC_ALLOW_TEST_BUILD="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "BREAKFLAG_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_soak_full_step_at_65536_envs():
    """1 500 full control steps (+ reset of finished envs) at the headline size under noisy actions (tools/soak.py, shortened): the state
    stays finite, nothing is launched, no hand-off of the dynamics kernel times out, episodes keep ending and restarting."""
    import torch
    from gpu_helpers import default_config
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 65536
    env = HipParkourEnv(default_config(), n, "cuda:0", False, seed=3, enable_dynamics=True, mirror_ref_state=False)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(17)
    env.reset()
    acts = [_bench_actions(env, gen) for _ in range(8)]
    ended = 0
    for it in range(1500):
        _, _, done, _ = env.step(acts[it & 7])
        if it % 250 == 249:
            _finite_state(env)
            assert float(env._char_root_vel.norm(dim=-1).max()) < 60.0 and float(env._char_contact_forces.abs().max()) < 2.0e5
            ended += int((done != 0).sum())
        env.reset_done()
    assert ended > 1000
    assert env.dynamics_timeouts() == 0
    assert env.dynamics_manifold_drops() == 0
    fr = env.get_fail_rates().numpy()
    assert np.isfinite(fr).all() and fr.min() > 0.0 and fr.max() <= 1.0


@pytest.mark.parametrize("kernel", ["wave", "coop", "thread"])
def test_out_of_range_actions_equal_host_clipped_actions_bit_for_bit(kernel):
    """SURVEY a23 on the device: `_apply_action` clips the action to the PD bounds before it becomes the joint targets
    (ig_char_env.py:488-490).  The bounds the env hands to the library equal the reference's own `_build_action_bounds_pd` (golden fixture
    action_bounds.npz, all 28 x 2 values, as float32), and a control step with actions far outside them equals, bit for bit, the step from
    the same state with the actions clipped ON THE HOST to the golden bounds -- for each of the three dynamics kernels."""
    import torch
    from conftest import golden
    from gpu_helpers import default_config
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    g = golden("action_bounds")
    n = 2048
    env = HipParkourEnv(default_config(), n, "cuda:0", False, seed=5, enable_dynamics=True, mirror_ref_state=False, dev_options={"kernel": kernel})
    assert env.describe()["dynamics_kernel"] == {"wave": "k_dynamics_wave", "coop": "k_dynamics_coop", "thread": "k_dynamics"}[kernel]
    lo32, hi32 = g["action_low"].astype(np.float32), g["action_high"].astype(np.float32)
    assert np.array_equal(env._action_bound_low.cpu().numpy(), lo32) and np.array_equal(env._action_bound_high.cpu().numpy(), hi32)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(9)
    env.reset()
    for _ in range(3):
        env.step(_bench_actions(env, gen)); env.reset_done()
    names = ("_char_root_pos", "_char_root_rot", "_char_root_vel", "_char_root_ang_vel", "_char_dof_pos", "_char_dof_vel", "_char_contact_forces",
             "_char_rigid_body_pos", "_timestep_buf", "_time_buf", "_motion_ids", "_motion_terrain_ids", "_motion_time_offsets")
    saved = {k: getattr(env, k).clone() for k in names}
    lo, hi = env._action_bound_low, env._action_bound_high
    raw = 0.5 * (hi + lo) + 3.0 * (hi - lo) * torch.randn(n, lo.shape[0], device="cuda:0", generator=gen)   # most entries far outside
    assert float(((raw < lo) | (raw > hi)).float().mean()) > 0.7
    clipped = torch.minimum(torch.maximum(raw, torch.from_numpy(lo32).to(raw.device)), torch.from_numpy(hi32).to(raw.device))
    out = []
    # (the first pass is a warm-up: only there does the root-position residual of the previous step still match the state -- after it
    # every pass starts from restored buffers, whose residual the kernel discards by value comparison)
    for act in (raw, raw, clipped):
        for k in names:
            getattr(env, k).copy_(saved[k])
        env.step(act.contiguous())
        torch.cuda.synchronize()
        out.append({k: getattr(env, k).clone() for k in names[:8]} | {"obs": env._obs_buf.clone(), "reward": env._reward_buf.clone()})
    for k in out[1]:
        assert torch.equal(out[1][k], out[2][k]), k
    assert torch.isfinite(out[1]["obs"]).all()


@pytest.mark.parametrize("mode", ["vel", "torque", "pd_exp"])
def test_control_modes_three_kernels_and_the_host_build_agree(mode):
    """The reference's control modes other than pd (ig_char_env.py:21-26, 374-421, 488-506): the wave kernel's own instantiation
    (k_dynamics_wave_ff), the chain-parallel and the thread-per-env kernel and the host build of the reference statement integrate the same
    control step from the same state; actions partly outside their bounds.  What the modes mean is checked on the host build
    (tests/test_dynamics_cpu.py::test_control_modes_vel_torque_pd_exp); PhysX parity is unpinned as for pd."""
    import torch
    from gpu_helpers import default_config, to_np
    from oracle.binding_dyn import DynOracle
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    n = 1024
    cfg = default_config(); cfg["env"]["control_mode"] = mode
    envs = {k: HipParkourEnv(cfg, n, "cuda:0", False, seed=21, enable_dynamics=True, mirror_ref_state=False, dev_options={"kernel": k}) for k in ("wave", "coop", "thread")}
    desc = envs["wave"].describe()
    assert desc["dynamics_kernel"] == "k_dynamics_wave_ff" and desc["control_mode"] == mode
    assert envs["coop"].describe()["dynamics_kernel"] == "k_dynamics_coop" and envs["thread"].describe()["dynamics_kernel"] == "k_dynamics"
    ref = envs["wave"]
    lo, hi = ref._action_bound_low, ref._action_bound_high
    if mode == "vel":
        assert float(hi.min()) == float(np.float32(2 * np.pi))
    for e in envs.values():
        e.reset()
    d = DynOracle(ref._scene.cfg)
    sc = ref._scene
    hf, mp, dxdy = sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy
    state = ["_char_root_pos", "_char_root_rot", "_char_root_vel", "_char_root_ang_vel", "_char_dof_pos", "_char_dof_vel"]
    tol = {"_char_root_pos": 1e-4, "_char_root_rot": 1e-4, "_char_root_vel": 2e-3, "_char_root_ang_vel": 1e-2, "_char_dof_pos": 2e-4, "_char_dof_vel": 5e-2}
    g = torch.Generator(device="cuda:0"); g.manual_seed(4)
    moved = 0.0
    # torque and pd_exp switch the joints' damping off (ig_char_env.py:124-129): a humanoid with undamped joints in stiff contact is chaotic within ONE
    # control step -- measured with zero torque (a limp character): every kernel within 1e-5 of the host build for the median env, q90 within the
    # tolerances below, and 1 % of the envs (feet flailing at 60 rad/s) apart by centimetres, all three kernels alike.  There the median and q90 are
    # the check; vel keeps the damping and is held to the tail like pd.
    damped = mode == "vel"
    for it in range(4):
        z = torch.randn(ref._char_dof_pos.shape, device="cuda:0", generator=g)
        if mode == "pd_exp":
            act = ref._char_dof_pos + 0.1 * z          # small pose errors: the explicit torque is held for a whole control step
        else:
            act = 0.5 * (hi + lo) + 0.03 * (hi - lo) * z  # vel: rad/s, torque: N m (gentle: undamped joints amplify every rounding difference)
        act = act.contiguous()
        for kern in ("coop", "thread"):
            for nm in state + ["_char_contact_forces"]:
                getattr(envs[kern], nm).copy_(getattr(ref, nm))
        st = dict(root_pos=to_np(ref._char_root_pos).copy(), root_rot=to_np(ref._char_root_rot).copy(), root_vel=to_np(ref._char_root_vel).copy(),
                  root_ang_vel=to_np(ref._char_root_ang_vel).copy(), dof_pos=to_np(ref._char_dof_pos).copy(), dof_vel=to_np(ref._char_dof_vel).copy(),
                  contact_force=np.zeros((n, 15, 3), np.float32))
        before = to_np(ref._char_dof_pos).copy()
        for e in envs.values():
            e.step(act)
        d.step(hf, mp, dxdy, st, to_np(act), sc.env_offsets)
        moved = max(moved, float(np.abs(to_np(ref._char_dof_pos) - before).max()))
        for kern in ("coop", "thread"):
            for nm in state:
                a, b = to_np(getattr(ref, nm)), to_np(getattr(envs[kern], nm))
                assert np.isfinite(a).all() and np.isfinite(b).all(), (it, kern, nm)
                err = np.abs(a - b).reshape(n, -1).max(1)
                if damped:
                    assert np.quantile(err, 0.99) <= 4 * tol[nm] and err.max() <= 50 * tol[nm], (it, kern, nm, np.quantile(err, 0.99), err.max())
                else:
                    assert np.median(err) <= 0.2 * tol[nm] and np.quantile(err, 0.9) <= 3 * tol[nm], (it, kern, nm, np.median(err), np.quantile(err, 0.9))
        for k_o, nm in [("root_pos", "_char_root_pos"), ("root_rot", "_char_root_rot"), ("root_vel", "_char_root_vel"), ("root_ang_vel", "_char_root_ang_vel"),
                        ("dof_pos", "_char_dof_pos"), ("dof_vel", "_char_dof_vel")]:
            err = np.abs(to_np(getattr(ref, nm)) - st[k_o]).reshape(n, -1).max(1)
            if damped:
                assert np.quantile(err, 0.99) <= 5 * tol[nm], (it, "host", nm, np.quantile(err, 0.99))
            else:
                assert np.median(err) <= 0.2 * tol[nm] and np.quantile(err, 0.9) <= 2 * tol[nm], (it, "host", nm, np.median(err), np.quantile(err, 0.9))
    assert moved > 0.02        # the actions did drive the joints
    if mode != "pd_exp":       # vel / torque clip the action to their bounds (ig_char_env.py:489-490): far-out actions = host-clipped actions, bit for bit
        names = state + ["_char_contact_forces", "_char_rigid_body_pos", "_timestep_buf", "_time_buf"]
        saved = {k: getattr(ref, k).clone() for k in names}
        raw = 0.5 * (hi + lo) + 3.0 * (hi - lo) * torch.randn(ref._char_dof_pos.shape, device="cuda:0", generator=g)
        out = []
        for act in (raw, raw, torch.minimum(torch.maximum(raw, lo), hi)):   # (first pass: warm-up of the root-position residual, as in the pd test below)
            for k in names:
                getattr(ref, k).copy_(saved[k])
            ref.step(act.contiguous()); torch.cuda.synchronize()
            out.append({k: getattr(ref, k).clone() for k in state})
        assert all(torch.equal(out[1][k], out[2][k]) for k in state)
    assert ref._lib.parc_env_dynamics_timeouts(ref._handle) == 0 and ref.dynamics_manifold_drops() == 0
    assert torch.isfinite(ref._obs_buf).all() and torch.isfinite(ref._reward_buf).all()
