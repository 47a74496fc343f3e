"""CPU tests of the host side: C-ABI export / struct layout, data-only .pkl reader, MJCF parser, scene assembly,
terrain builder, config handling.  No compute call needs a GPU."""
import ctypes as C
import os
import pickle
import re

import numpy as np
import pytest

from conftest import DATA, REPO, golden
from parc_amd import lib as L
from parc_amd import ms_file, terrain
from parc_amd.char_model import CharModel, JointType
from parc_amd.envs import scene
from parc_amd.util import path_loader


def default_config():
    return path_loader.load_config(os.path.join(DATA, "configs/tracker_config/dm_env_default.yaml"))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "parc_env.h")).read()
    declared = set(re.findall(r"\b(parc_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    so = C.CDLL(L.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(so, sym), f"{sym} declared in include/parc_env.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS)
    assert so.parc_abi_version() == L.ABI_VERSION
    # and nothing undeclared leaks out of the shipped library
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("parc_")}
    assert exported == declared, (exported - declared, declared - exported)


def test_struct_layout_matches_c_and_create_fails_loudly_without_gpu():
    """parc_env_create validates abi_version/struct_size before touching the device: the ctypes mirror must have the
    C layout.  Without a GPU the call must fail (no silent CPU path)."""
    lib = L.load()
    sc = scene.build_scene(default_config(), 8, verbose=False)
    h = C.c_void_p()
    bad = L.ParcEnvConfig.from_buffer_copy(sc.cfg)
    bad.struct_size = C.sizeof(L.ParcEnvConfig) - 4
    assert lib.parc_env_create(C.byref(bad), C.byref(h)) == -1
    assert b"ABI mismatch" in lib.parc_last_error()
    rc = lib.parc_env_create(C.byref(sc.cfg), C.byref(h))
    import torch
    if not torch.cuda.is_available():
        assert rc == -4 and b"HIP device" in lib.parc_last_error()   # PARC_ERR_NO_DEVICE, not a fallback
        from parc_amd.envs.hip_parkour_env import HipParkourEnv
        with pytest.raises(RuntimeError):
            HipParkourEnv(default_config(), 8, "cuda:0", False)
    else:
        assert rc == 0
        lib.parc_env_destroy(h)


def test_ms_file_reader_is_data_only(tmp_path):
    d = ms_file.load_ms_file(os.path.join(DATA, "motion_terrains", "sfu.pkl"))
    assert d.motion_data.root_pos.shape == (15, 3) and d.motion_data.fps == 30 and d.motion_data.loop_mode == "CLAMP"
    assert d.terrain_data.hf.shape == (32, 32) and abs(d.terrain_data.dx - 0.4) < 1e-6

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /tmp/parc_pwned",))
    for payload in (pickle.dumps(Evil()), pickle.dumps({"a": Evil()}), pickle.dumps(np.array([Evil()], dtype=object))):
        with pytest.raises(ms_file.UnsafePickleError):
            ms_file.loads_data_only(payload)
    assert not os.path.exists("/tmp/parc_pwned")
    # round trip through the reference-compatible writer
    out = tmp_path / "x.pkl"
    ms_file.save_ms_file(d, str(out))
    d2 = ms_file.load_ms_file(str(out))
    assert np.array_equal(d2.motion_data.joint_rot, d.motion_data.joint_rot) and np.array_equal(d2.terrain_data.hf, d.terrain_data.hf)
    # container layout of the reference (file_io.py:87-103): outer dict of three independently pickled payloads
    outer = pickle.load(open(out, "rb"))
    assert set(outer.keys()) == {"motion_data", "terrain_data", "misc_data"} and isinstance(outer["motion_data"], bytes)


def test_char_model_matches_reference_tables():
    g = golden("char_model")
    cm = CharModel(os.path.join(DATA, "assets", "humanoid.xml"))
    assert cm.get_body_names() == [str(n) for n in g["body_names"]]
    assert np.array_equal(cm._parent_indices, g["parent"])
    assert np.array_equal(cm._local_translation, g["local_translation"]) and np.array_equal(cm._local_rotation, g["local_rotation"])
    assert np.array_equal(cm.joint_type_array(), g["joint_type"]) and np.array_equal(cm.dof_idx_array(), g["dof_idx"])
    assert np.array_equal(cm.joint_axis_array(), g["joint_axis"]) and cm.get_dof_size() == int(g["dof_size"])
    lo, hi = cm.dof_limits()
    assert np.allclose(lo, g["lower"], atol=1e-7) and np.allclose(hi, g["upper"], atol=1e-7)
    paths = cm.fk_paths()
    assert [list(p[p >= 0]) for p in paths[:5]] == [[1, 2], [1, 3, 4, 5], [1, 6, 7, 8], [9, 10, 11], [12, 13, 14]]
    kp, kd, arm, eff = cm.dof_pd_params()
    assert kp[0] == 1000 and kd[0] == 100 and abs(arm[0] - 0.02) < 1e-7 and eff[0] == 200 and eff[9] == 70


def test_scene_assembly_and_action_bounds():
    sc = scene.build_scene(default_config(), 16, verbose=False)
    assert sc.cfg.num_rays == 441 and sc.cfg.num_tar_obs_steps == 6 and sc.key_body_ids == [5, 8, 11, 14]
    assert sum(int(np.prod(v["shape"])) for v in sc.obs_shapes.values()) == 1312
    # spherical: +-1.2 max|limit|; hinge: mid +- 0.7 range (ig_char_env.py:307-347)
    assert np.isclose(sc.action_high[0], 1.2 * np.deg2rad(90)) and np.isclose(sc.action_low[0], -1.2 * np.deg2rad(90))
    assert np.isclose(sc.action_low[9], np.deg2rad(80) - 0.7 * np.deg2rad(160)) and np.isclose(sc.action_high[9], np.deg2rad(80) + 0.7 * np.deg2rad(160))
    # env origins: ig_parkour_env.py:389-398; a shard uses its global env indices
    off = scene.env_offsets_square(16, 2.0)
    assert np.array_equal(off[5], [4.0, 4.0, 0.0])
    sh = scene.env_offsets_square(8, 2.0, env_id_base=8, total_envs=16)
    assert np.array_equal(sh, off[8:16])
    cfg = default_config(); cfg["env"]["global_obs"] = True   # built since round 3 (k_env_post<..., GLOBALOBS>)
    assert scene.build_scene(cfg, 4, verbose=False).cfg.global_obs == 1
    cfg = default_config(); cfg["env"]["global_root_height_obs"] = True   # one more observation in front (built since round 3)
    sc1 = scene.build_scene(cfg, 4, verbose=False)
    assert sc1.cfg.global_root_height_obs == 1 and sc1.obs_shapes["char_obs"]["shape"] == (137,)
    cfg = default_config(); cfg["env"]["use_contact_info"] = False          # built since round 4: no contact blocks (ig_parkour_env.py:927-946)
    sc2 = scene.build_scene(cfg, 4, verbose=False)
    assert sc2.cfg.use_contact_info == 0 and list(sc2.obs_shapes) == ["char_obs", "tar_obs", "hf"]
    cfg["env"]["enable_tar_obs"] = False
    sc3 = scene.build_scene(cfg, 4, verbose=False)
    assert sc3.cfg.enable_tar_obs == 0 and list(sc3.obs_shapes) == ["char_obs", "hf"] and sum(int(np.prod(v["shape"])) for v in sc3.obs_shapes.values()) == 136 + 441
    # control modes (ig_char_env.py:21-26, 252-268, 349-362): built since round 4 -- their action bounds here, their dynamics in test_dynamics_*
    cfg = default_config(); cfg["env"]["control_mode"] = "vel"
    sv = scene.build_scene(cfg, 4, verbose=False)
    assert sv.cfg.dynamics.control_mode == 1 and np.allclose(sv.action_low, -2 * np.pi) and np.allclose(sv.action_high, 2 * np.pi)
    cfg["env"]["control_mode"] = "torque"
    st = scene.build_scene(cfg, 4, verbose=False)
    eff = np.array([st.cfg.dynamics.dof_effort[d] for d in range(len(st.action_low))])
    assert st.cfg.dynamics.control_mode == 2 and eff.min() > 0 and np.allclose(st.action_high, eff) and np.allclose(st.action_low, -eff)
    cfg["env"]["control_mode"] = "pd_exp"
    se = scene.build_scene(cfg, 4, verbose=False)
    assert se.cfg.dynamics.control_mode == 3 and np.array_equal(se.action_low, sc.action_low) and np.array_equal(se.action_high, sc.action_high)
    cfg["env"]["control_mode"] = "pd_1d"                                     # the humanoid has spherical joints: the reference asserts (:246-250)
    with pytest.raises(AssertionError):
        scene.build_scene(cfg, 4, verbose=False)
    cfg["env"]["control_mode"] = "pid"
    with pytest.raises(KeyError):                                           # ControlMode["pid"]
        scene.build_scene(cfg, 4, verbose=False)
    # developer switches travel in the config (ParcEnvConfig.dev_options), the library reads no environment variable
    assert scene.format_dev_options(None) is None and scene.format_dev_options({"kernel": "wave"}) is None
    assert scene.format_dev_options({"segments": "none", "dtang": 1e4}) == b"dtang=10000.0;segments=none"
    assert scene.format_dev_options("kernel=coop; man_period=1") == b"kernel=coop;man_period=1"
    cfg = default_config(); del cfg["env"]["pose_w"]
    with pytest.raises(KeyError):   # required key, like the reference
        scene.build_scene(cfg, 4, verbose=False)


def test_terrain_square_matches_reference_and_cache_roundtrip(tmp_path):
    from helpers import load_clips
    g = golden("env_step")
    clips = load_clips([str(c) for c in g["clips"]])
    subs = []
    for c in clips:
        t = terrain.SubTerrain(c["hf"].shape[0], c["hf"].shape[1], c["dx"], c["dx"], c["min_point"][0], c["min_point"][1])
        t.hf = c["hf"].copy(); subs.append(t)
    grid = terrain.build_terrain_square(subs, 0.4, 0.4)
    assert np.array_equal(grid.terrain.hf, g["hf"]) and np.array_equal(grid.terrain.min_point, g["hf_min_point"])
    assert np.array_equal(grid.motion_offsets, g["motion_offsets"])
    p = str(tmp_path / "terrain.pkl")
    terrain.save_terrain(grid, p)
    back = terrain.load_terrain(p)
    assert np.array_equal(back.terrain.hf, grid.terrain.hf) and np.array_equal(back.motion_offsets, grid.motion_offsets)
    ray = terrain.get_xy_points_cone(0.05, 2, 60, 3, 3, 0.26179938779)
    assert np.abs(ray - g["ray_points"]).max() < 5e-7
    tl = golden("terrain_lookup")
    t = terrain.SubTerrain(102, 102, 0.4, 0.4, tl["min_point"][0], tl["min_point"][1]); t.hf = tl["hf"]
    assert np.array_equal(t.get_grid_index(tl["points"]), tl["grid_index"])


def test_path_loader_data_dir(monkeypatch, tmp_path):
    monkeypatch.setenv("PARC_DATA_DIR", str(tmp_path))
    assert str(path_loader.resolve_path("$DATA_DIR/a/b.yaml")) == str(tmp_path / "a" / "b.yaml")
    monkeypatch.delenv("PARC_DATA_DIR")
    assert str(path_loader.resolve_path("$DATA_DIR/assets/humanoid.xml")).endswith("data/assets/humanoid.xml")


def test_terrain_slice_matches_reference():
    """slice_terrain_around_motion as the recorder uses it (terrain_util.py:1587-1642, ig_parkour_env.py:715-720)."""
    from conftest import golden
    from parc_amd import terrain as T
    g = golden("terrain_slice")
    t = T.SubTerrain(g["hf"].shape[0], g["hf"].shape[1], g["dxdy"][0], g["dxdy"][1], g["min_point"][0], g["min_point"][1])
    t.hf = g["hf"].astype(np.float32); t.hf_maxmin = g["hf_maxmin"].astype(np.float32)
    for i in range(3):
        st, loc = T.slice_terrain_around_motion(g[f"frames{i}"], t, padding=float(g[f"padding{i}"]))
        assert tuple(st.hf.shape) == tuple(g[f"hf{i}"].shape) == tuple(int(x) for x in g[f"dims{i}"])
        np.testing.assert_array_equal(st.hf, g[f"hf{i}"])
        np.testing.assert_array_equal(st.hf_maxmin, g[f"hf_maxmin{i}"])
        np.testing.assert_array_equal(st.min_point, g[f"min_point{i}"])
        np.testing.assert_array_equal(loc, g[f"local{i}"])


def test_build_terrain_wide_matches_reference():
    """terrain_build_mode: wide (dm_env.py:318-445) against the reference's own output, 4 clips x 2 copies."""
    import helpers
    g = golden("terrain_wide")
    subs = [terrain.SubTerrain.from_ms_terrain_data(ms_file.load_ms_file(helpers.clip_path(str(c)), load_misc=False).terrain_data)
            for c in g["clips"]]
    grid = terrain.build_terrain_wide(subs, 0.4, 0.4, 2)
    assert grid.terrains_per_motion == 2
    np.testing.assert_array_equal(grid.terrain.dims, g["dims"])
    np.testing.assert_array_equal(grid.terrain.hf, g["hf"])
    np.testing.assert_array_equal(grid.terrain.min_point, g["min_point"])
    np.testing.assert_array_equal(grid.motion_offsets, g["motion_offsets"])


def test_ray_division_shortcut_is_exact():
    """cell_index_rcp (parc_env.hip) == IEEE division for the grid spacings in use: every float numerator between 2^-7 and
    2^13 metres, both signs (the ray loop's numerators are point - terrain min, within +-8 km)."""
    import subprocess
    here = os.path.join(REPO, "oracle")
    subprocess.check_call(["make", "-C", here, "-s", "libdiv_check.so"])
    lib = C.CDLL(os.path.join(here, "libdiv_check.so"))
    lib.parc_check_rcp_division.restype = C.c_longlong
    lib.parc_check_rcp_division.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    for d in (0.4, 0.1, 0.25, 0.3):
        bad_x = C.c_float(0.0)
        lo, hi = (2.0 ** -7, 2.0 ** 13) if d == 0.4 else (2.0 ** -3, 2.0 ** 9)
        n = lib.parc_check_rcp_division(np.float32(d), lo, hi, C.byref(bad_x))
        assert n == 0, (d, n, bad_x.value)


def test_create_dataset_yaml_class_balanced_weights(tmp_path):
    """util/create_dataset.py:20-177: weight = length x factor with equal total mass per first-level folder; the YAML loads
    back through the motion-file loader."""
    import shutil
    import yaml
    from parc_amd import motion_lib
    from parc_amd.util.create_dataset import create_dataset_yaml
    root = tmp_path / "ds"
    for cls, clips in {"walk": ["civilization", "sfu"], "parkour": ["TEASER_TERRAIN"], "ignore_me": ["sfu"]}.items():
        (root / cls / "batch0").mkdir(parents=True)
        for c in clips:
            shutil.copy(os.path.join(DATA, "motion_terrains", c + ".pkl"), root / cls / "batch0" / (c + "_" + cls + ".pkl"))
    out = create_dataset_yaml([root], tmp_path / "ds.yaml", verbose=False)
    y = yaml.safe_load(open(tmp_path / "ds.yaml"))
    assert len(y["motions"]) == 3 and not any("ignore" in m["file"] for m in y["motions"])
    mass = {}
    for m in y["motions"]:
        cls = "walk" if "/walk/" in m["file"] else "parkour"
        mass[cls] = mass.get(cls, 0.0) + m["weight"]
    assert abs(mass["walk"] - mass["parkour"]) < 1e-9 * max(mass.values())
    w = {os.path.basename(m["file"]): m["weight"] for m in y["motions"]}
    assert abs(w["civilization_walk.pkl"] / w["sfu_walk.pkl"] - (254 / 30) / (15 / 30)) < 1e-9   # within a class: proportional to length
    clips = motion_lib.load_motion_file(str(tmp_path / "ds.yaml"), verbose=False)
    assert [c.name for c in clips] == [os.path.splitext(os.path.basename(m["file"]))[0] for m in y["motions"]]


def _device_poly(fn):
    """Coefficients of a Horner polynomial of parc_math.hpp, in evaluation order (highest degree first)."""
    src = open(os.path.join(REPO, "parc_amd", "csrc", "parc_math.hpp")).read()
    body = src[src.index("float " + fn + "("):]
    body = body[:body.index("\n}\n")]
    first = re.search(r"float p = (-?[0-9.e+-]+)f;", body).group(1)
    rest = re.findall(r"p = fmaf\(p, z, (-?[0-9.e+-]+)f\);", body)
    return [np.float32(first)] + [np.float32(c) for c in rest]


def _fma32(a, b, c):  # fl32(a * b + c): the fp32 product is exact in fp64
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_reduced_range_polynomials_of_the_observation_slerp():
    """sin on [0, pi/2] and acos on [0, 1] as slerp_rr evaluates them (parc_math.hpp: same coefficients, same Horner order, fp32 with
    fused multiply-adds), against the exact functions on dense grids: <= 2e-7 relative, i.e. the accuracy class of the library
    routines they stand in for.  (The device result itself is checked by the GPU tests.)"""
    cs = _device_poly("sin_q1")
    x = np.linspace(0.0, 1.62, 3_000_001).astype(np.float32)
    z = x * x
    p = np.full_like(z, cs[0])
    for c in cs[1:]:
        p = _fma32(p, z, np.full_like(z, c))
    s = _fma32(x * z, p, x)
    ref = np.sin(x.astype(np.float64))
    assert np.max(np.abs(s - ref)[1:] / ref[1:]) <= 2e-7
    ca = _device_poly("acos_01")
    c = np.linspace(0.0, 1.0, 4_000_001).astype(np.float32)
    big = c > np.float32(0.5)
    zb = (np.float32(1.0) - c) * np.float32(0.5)
    z = np.where(big, zb, c * c).astype(np.float32)
    u = np.where(big, np.sqrt(zb), c).astype(np.float32)
    p = np.full_like(z, ca[0])
    for q in ca[1:]:
        p = _fma32(p, z, np.full_like(z, q))
    a = _fma32(u * z, p, u)
    h = np.where(big, np.float32(2.0) * a, np.float32(1.5707963267948966) - a).astype(np.float32)
    ref = np.arccos(c.astype(np.float64))
    err = np.abs(h - ref)
    assert err.max() <= 2e-7
    far = ref > 1e-4  # relative accuracy is what the sin ratios need near c -> 1
    assert np.max(err[far] / ref[far]) <= 2e-7


def test_read_motion_data_script_prints_the_clip(capsys):
    """scripts/read_motion_data.py (BASELINE cfg 1's plumbing script, reference scripts/read_motion_data.py:1-20): same five prints,
    read through the data-only decoder; values against SURVEY 8(c)'s loader sanity (sfu.pkl: 15 frames, 30 fps)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("read_motion_data", os.path.join(REPO, "scripts", "read_motion_data.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = mod.main(["read_motion_data.py"])
    out = capsys.readouterr().out
    md = d.motion_data
    assert md.root_pos.shape == (15, 3) and md.root_rot.shape == (15, 4) and md.joint_rot.shape == (15, 14, 4) and md.body_contacts.shape == (15, 15)
    assert md.fps == 30 and out.strip().splitlines()[-1] == "30"
    assert str(md.root_pos) in out and str(md.body_contacts) in out


def test_quat_to_exp_map_host_helper_matches_reference_rows():
    """The legacy recorder format converts the recorded root quaternion with the reference's quat_to_exp_map (torch_util.py:372):
    the numpy helper against the reference's own rows (quat_ops.npz, incl. w < 0 and tiny angles)."""
    from parc_amd.envs.hip_parkour_env import _quat_to_exp_map_np
    g = golden("quat_ops")
    np.testing.assert_allclose(_quat_to_exp_map_np(g["a"]), g["quat_to_exp_map"], atol=2e-6, rtol=0)


def test_dynamics_kernel_of_the_built_library_uses_no_scratch(tmp_path):
    """k_dynamics_wave keeps every body of a chain in registers: that needs its body loops fully unrolled (build flag
    -pragma-unroll-threshold, __graft_entry__.py) -- with the loops rolled the per-body arrays turn into ~1 KB of scratch per lane and the
    kernel is several times slower, silently.  Read from the code object of the library as built: private segment 0, 1 wave per SIMD."""
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("LLVM binutils of ROCm not present")
    so = tmp_path / "lib.so"
    shutil.copy(L.LIB_PATH, so)
    subprocess.check_call([objdump, "--offloading", str(so)], cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(co) == 1, os.listdir(tmp_path)
    notes = subprocess.check_output([readelf, "--notes", str(tmp_path / co[0])], text=True)
    kern = {}
    for blk in notes.split("- .agpr_count:")[1:]:     # one metadata block per kernel
        m = re.search(r"\.name:\s+(\S+)", blk)
        if m:
            kern[m.group(1)] = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|sgpr_count|group_segment_fixed_size):\s+(\d+)", blk)}
    wave = [v for k, v in kern.items() if "k_dynamics_wave" in k]
    assert len(wave) == 2, list(kern)                   # the pd instantiation and k_dynamics_wave_ff (the other control modes)
    for wk in wave:
        assert wk["private_segment_fixed_size"] == 0, wk
        assert wk["vgpr_count"] > 256, wk               # VGPRs + AGPRs of the one resident wave per SIMD
    post = [v for k, v in kern.items() if "k_env_post" in k]
    assert len(post) == 9 and all(p["vgpr_count"] <= 102 for p in post), post   # (STEP, OBS) x MIRROR, the two track_root=false STEP instantiations, three with global_obs; 5 waves per SIMD need <= 102 registers
