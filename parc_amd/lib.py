"""ctypes binding of ``libparc_env.so`` (C-ABI in ``include/parc_env.h``).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the env raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PARC_ENV_LIB: developer override used to A/B kernel builds; the shipped library is the in-tree one
LIB_PATH = os.environ.get("PARC_ENV_LIB") or os.path.join(_HERE, "libparc_env.so")

ABI_VERSION = 6
MAX_BODIES, MAX_DOFS, MAX_TAR_STEPS, MAX_KEY, MAX_FK_PATHS, MAX_FK_DEPTH, MAX_GEOMS = 16, 40, 6, 8, 8, 8, 24

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)


class ParcCharModel(C.Structure):
    _fields_ = [("num_bodies", C.c_int32), ("dof_size", C.c_int32), ("parent", C.c_int32 * MAX_BODIES),
                ("local_translation", (C.c_float * 3) * MAX_BODIES), ("local_rotation", (C.c_float * 4) * MAX_BODIES),
                ("joint_type", C.c_int32 * MAX_BODIES), ("joint_axis", (C.c_float * 3) * MAX_BODIES),
                ("dof_idx", C.c_int32 * MAX_BODIES), ("fk_paths", (C.c_int32 * MAX_FK_DEPTH) * MAX_FK_PATHS)]


class ParcDynamicsParams(C.Structure):
    _fields_ = [("num_geoms", C.c_int32), ("geom_body", C.c_int32 * MAX_GEOMS), ("geom_type", C.c_int32 * MAX_GEOMS),
                ("geom_pos", (C.c_float * 3) * MAX_GEOMS), ("geom_pos2", (C.c_float * 3) * MAX_GEOMS),
                ("geom_size", (C.c_float * 3) * MAX_GEOMS), ("geom_density", C.c_float * MAX_GEOMS),
                ("dof_stiffness", C.c_float * MAX_DOFS), ("dof_damping", C.c_float * MAX_DOFS),
                ("dof_armature", C.c_float * MAX_DOFS), ("dof_effort", C.c_float * MAX_DOFS),
                ("dof_lower", C.c_float * MAX_DOFS), ("dof_upper", C.c_float * MAX_DOFS),
                ("gravity_z", C.c_float), ("sim_dt", C.c_float), ("sim_steps", C.c_int32), ("substeps", C.c_int32),
                ("solver_iterations", C.c_int32), ("friction", C.c_float), ("restitution", C.c_float),
                ("contact_offset", C.c_float), ("max_depenetration_velocity", C.c_float),
                ("angular_damping", C.c_float), ("max_angular_velocity", C.c_float), ("control_mode", C.c_int32)]


class ParcEnvConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("struct_size", C.c_uint32), ("device", C.c_int32), ("num_envs", C.c_int32),
                ("model", ParcCharModel), ("num_key_bodies", C.c_int32), ("key_body_ids", C.c_int32 * MAX_KEY),
                ("num_tar_obs_steps", C.c_int32), ("tar_obs_steps", C.c_int32 * MAX_TAR_STEPS),
                ("num_rays", C.c_int32), ("ray_points_host", f32p), ("control_dt", C.c_double),
                ("episode_length", C.c_float), ("min_obs_h", C.c_float), ("max_obs_h", C.c_float),
                ("pose_w", C.c_float), ("vel_w", C.c_float), ("root_pos_w", C.c_float), ("root_vel_w", C.c_float),
                ("key_pos_w", C.c_float), ("joint_err_w", C.c_float * MAX_BODIES), ("dof_err_w", C.c_float * MAX_DOFS),
                ("contact_weights", C.c_float * MAX_BODIES), ("pose_termination_dist", C.c_float * MAX_BODIES),
                ("root_pos_termination_dist", C.c_float), ("root_rot_termination_angle", C.c_float),
                ("enable_early_termination", C.c_int32), ("pose_termination", C.c_int32), ("track_root", C.c_int32),
                ("track_root_h", C.c_int32), ("report_tracking_error", C.c_int32),
                ("fail_rate_ema_weight", C.c_float), ("min_motion_weight", C.c_float),
                ("rand_root_pos_offset_scale", C.c_float), ("rand_reset", C.c_int32), ("demo_mode", C.c_int32),
                ("env_offsets_host", f32p), ("action_low", C.c_float * MAX_DOFS), ("action_high", C.c_float * MAX_DOFS),
                ("body_pos_from_fk", C.c_int32), ("enable_dynamics", C.c_int32), ("dynamics", ParcDynamicsParams),
                ("seed", C.c_uint64), ("contact_body_mask", C.c_uint32), ("termination_height", C.c_float),
                ("global_obs", C.c_int32), ("global_root_height_obs", C.c_int32),
                ("use_contact_info", C.c_int32), ("enable_tar_obs", C.c_int32), ("dev_options", C.c_char_p)]


class ParcMotionClips(C.Structure):
    _fields_ = [("num_motions", C.c_int32), ("num_frames_host", i32p), ("fps_host", i32p), ("loop_modes_host", i32p),
                ("weights_host", f64p), ("root_pos_host", f32p), ("root_rot_host", f32p), ("joint_rot_host", f32p),
                ("contacts_host", f32p)]


BUFFER_FIELDS = [
    ("char_root_pos", "f"), ("char_root_rot", "f"), ("char_root_vel", "f"), ("char_root_ang_vel", "f"),
    ("char_dof_pos", "f"), ("char_dof_vel", "f"), ("char_body_pos", "f"), ("contact_forces", "f"),
    ("motion_ids", "i"), ("terrain_ids", "i"), ("time_offsets", "f"), ("timestep", "i"), ("time", "f"), ("ep_num", "l"),
    ("obs", "f"), ("reward", "f"), ("done", "i"), ("reward_terms", "f"), ("tracking_error", "f"),
    ("ref_root_pos", "f"), ("ref_root_rot", "f"), ("ref_root_vel", "f"), ("ref_root_ang_vel", "f"),
    ("ref_joint_rot", "f"), ("ref_dof_pos", "f"), ("ref_dof_vel", "f"), ("ref_body_pos", "f"), ("ref_contacts", "f"),
    ("ray_hfs", "f"),
]


class ParcEnvBuffers(C.Structure):
    _fields_ = [(n, {"f": f32p, "i": i32p, "l": i64p}[t]) for n, t in BUFFER_FIELDS]


_lib = None


def load():
    """Load the HIP library; raises if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback for the env step.")
    import torch  # noqa: F401  the process must hold ONE HIP runtime (torch's): load it before the library binds to libamdhip64
    lib = C.CDLL(LIB_PATH)
    stale = f"{LIB_PATH} is a stale build (ABI mismatch): rebuild with `python -c 'import __graft_entry__ as g; g.build()'`"
    if not hasattr(lib, "parc_abi_version") or lib.parc_abi_version() != ABI_VERSION or any(not hasattr(lib, s) for s in EXPORTED_SYMBOLS):
        raise RuntimeError(stale)
    lib.parc_build_flags.restype = C.c_char_p
    flags = lib.parc_build_flags().decode()
    missing = [f for f in REQUIRED_BUILD_FLAGS if f not in flags.split()]
    if any(m in flags for m in TEST_BUILD_MARKERS) and os.environ.get("PARC_ALLOW_TEST_BUILD") != "1":
        raise RuntimeError(f"{LIB_PATH} is a test-only build ('{flags}'): the product never loads it")
    if missing:  # an SLP-vectorised k_dynamics_wave computes wrong inertias; contraction breaks the 1e-5 parity
        raise RuntimeError(f"{LIB_PATH} was built with '{flags}': missing {missing}; rebuild with __graft_entry__.build()")
    lib.parc_last_error.restype = C.c_char_p
    vp = C.c_void_p
    lib.parc_env_create.argtypes = [C.POINTER(ParcEnvConfig), C.POINTER(vp)]
    lib.parc_env_destroy.argtypes = [vp]
    lib.parc_env_destroy.restype = None
    lib.parc_env_obs_dim.argtypes = [vp]
    lib.parc_env_load_motions.argtypes = [vp, C.POINTER(ParcMotionClips)]
    lib.parc_env_load_terrain.argtypes = [vp, f32p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                          f32p, C.c_int32, C.c_int32]
    lib.parc_env_bind_buffers.argtypes = [vp, C.POINTER(ParcEnvBuffers)]
    lib.parc_env_step.argtypes = [vp, vp, vp]
    lib.parc_env_reset.argtypes = [vp, vp, C.c_int32, vp]
    lib.parc_env_reset_with.argtypes = [vp, vp, C.c_int32, vp, vp, vp, vp, vp]
    lib.parc_env_compute_obs.argtypes = [vp, vp, C.c_int32, vp]
    lib.parc_env_reset_done.argtypes = [vp, vp]
    lib.parc_env_get_fail_rates.argtypes = [vp, f32p, C.c_int32]
    lib.parc_env_set_fail_rates.argtypes = [vp, f32p, C.c_int32]
    lib.parc_env_get_motion_info.argtypes = [vp, f32p, f32p, C.c_int32]
    lib.parc_env_set_rand_reset.argtypes = [vp, C.c_int32, C.c_int32, C.c_float]
    lib.parc_env_set_start_time_fraction.argtypes = [vp, vp]
    lib.parc_dof_to_rot.argtypes = [vp, vp, vp, C.c_int32, vp]
    lib.parc_rot_to_dof.argtypes = [vp, vp, vp, C.c_int32, vp]
    lib.parc_forward_kinematics.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, vp]
    lib.parc_calc_motion_frame.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.parc_env_get_frame_vel_tables.argtypes = [vp, f32p, f32p, f32p]
    lib.parc_env_profile_step.argtypes = [vp, vp, vp, C.c_int32, f32p, f32p]
    lib.parc_env_last_dynamics_ms.argtypes = [vp]
    lib.parc_env_last_dynamics_ms.restype = C.c_float
    lib.parc_env_get_buffers.argtypes = [vp, C.POINTER(ParcEnvBuffers)]
    lib.parc_env_bind_action.argtypes = [vp, vp]
    lib.parc_env_step_reset_graph.argtypes = [vp, vp]
    lib.parc_normalize_record.argtypes = [vp, vp, vp, C.c_float, vp, vp, C.c_int64, C.c_int32, vp]
    lib.parc_td_lambda_return.argtypes = [vp, vp, vp, C.c_float, C.c_float, C.c_int32, C.c_int32, vp, vp]
    lib.parc_env_set_episode_length.argtypes = [vp, C.c_float]
    lib.parc_env_record_bind.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp, C.c_int32]
    lib.parc_env_record_frame.argtypes = [vp, vp]
    lib.parc_env_set_kernel_timing.argtypes = [vp, C.c_int32]
    lib.parc_env_get_kernel_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.parc_env_dynamics_kernel.argtypes = [vp]
    lib.parc_env_set_never_done.argtypes = [vp, C.c_int32]
    lib.parc_env_dynamics_timeouts.argtypes = [vp]
    lib.parc_env_describe.argtypes = [vp]
    lib.parc_gather_rows.argtypes = [C.c_int32, C.POINTER(vp), C.POINTER(vp), i64p, vp, C.c_int64, C.c_int64, vp]
    lib.parc_env_health_words.argtypes = [vp]
    lib.parc_env_health_words.restype = C.POINTER(C.c_uint32)
    lib.parc_env_get_kernel_timing_samples.argtypes = [vp, f32p, f32p, f32p, C.c_int32, C.POINTER(C.c_int32)]
    lib.parc_env_describe.restype = C.c_char_p
    lib.parc_env_dynamics_manifold_drops.argtypes = [vp]
    lib.parc_test_quat_op.argtypes = [C.c_int32, vp, vp, vp, C.c_int32, vp, vp]
    lib.parc_env_dynamics_kernel.restype = C.c_char_p
    lib.parc_env_post_kernel.argtypes = [vp]
    lib.parc_env_post_kernel.restype = C.c_char_p
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "parc_last_error", "parc_abi_version", "parc_env_create", "parc_env_destroy", "parc_env_obs_dim",
    "parc_env_load_motions", "parc_env_load_terrain", "parc_env_bind_buffers", "parc_env_step", "parc_env_reset",
    "parc_env_reset_with", "parc_env_reset_done", "parc_env_compute_obs", "parc_env_get_fail_rates", "parc_env_set_fail_rates",
    "parc_env_get_motion_info", "parc_env_set_rand_reset", "parc_env_set_start_time_fraction", "parc_dof_to_rot",
    "parc_rot_to_dof", "parc_forward_kinematics", "parc_calc_motion_frame", "parc_env_get_frame_vel_tables",
    "parc_env_profile_step", "parc_env_last_dynamics_ms", "parc_env_dynamics_kernel", "parc_env_set_kernel_timing", "parc_env_get_kernel_timing", "parc_env_record_bind", "parc_env_record_frame", "parc_env_set_episode_length", "parc_td_lambda_return", "parc_normalize_record", "parc_env_bind_action", "parc_env_get_buffers", "parc_env_step_reset_graph",
    "parc_test_quat_op", "parc_build_flags", "parc_env_set_never_done", "parc_env_dynamics_timeouts", "parc_env_dynamics_manifold_drops", "parc_env_describe", "parc_env_get_kernel_timing_samples", "parc_env_health_words", "parc_gather_rows",
    "parc_env_post_kernel",
]

# parc_test_quat_op selectors (include/parc_env.h)
QOP = {"mul": 0, "rotate": 1, "conj": 2, "pos": 3, "normalize3": 4, "to_axis_angle": 5, "aa_to_quat": 6, "exp_map_to_quat": 7,
       "to_exp_map": 8, "diff_angle": 9, "normalize": 10, "to_tan_norm": 11, "slerp": 12, "heading": 13, "heading_quat_inv": 14,
       "diff": 15, "rotate_2d": 16, "slerp_rr": 17}
# -pragma-unroll-threshold: without it the unrolled body loops of k_dynamics_wave stay rolled, its per-body register arrays become ~1 KB of
# scratch per lane and the kernel is several times slower -- a library built without it is refused like one built with SLP
REQUIRED_BUILD_FLAGS = ("-fno-slp-vectorize", "-ffp-contract=off", "-pragma-unroll-threshold=1048576")
# defines of TEST-ONLY builds (__graft_entry__.BREAK_FLAGS): load() refuses a library whose build-flag string carries one unless the
# caller opts in (PARC_ALLOW_TEST_BUILD=1, set by the one test that loads the break-flag variant)
TEST_BUILD_MARKERS = ("-DPARC_TEST_BREAK_FLAG",)


class ParcError(RuntimeError):
    pass


def csrc_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources + the C-ABI header: profiles/*.json carry it, so that bench.py attaches
    counter-derived figures only to the build they were collected on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for f in sorted(os.listdir(csrc)) + [os.path.join("..", "..", "include", "parc_env.h")]:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def check(rc):
    if rc != 0:
        raise ParcError(f"libparc_env error {rc}: {load().parc_last_error().decode()}")


def np_f32p(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(f32p)


def np_i32p(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(i32p)


def make_char_model(cm) -> ParcCharModel:
    """Pack a :class:`parc_amd.char_model.CharModel` into the C struct."""
    m = ParcCharModel()
    nb = cm.get_num_bodies()
    if nb > 15:
        raise ValueError("at most 15 bodies are supported by the gfx950 lane map")
    m.num_bodies = nb
    m.dof_size = cm.get_dof_size()
    jt, ax, di = cm.joint_type_array(), cm.joint_axis_array(), cm.dof_idx_array()
    for b in range(nb):
        m.parent[b] = int(cm._parent_indices[b])
        m.joint_type[b] = int(jt[b])
        m.dof_idx[b] = int(di[b])
        for k in range(3):
            m.local_translation[b][k] = float(cm._local_translation[b][k])
            m.joint_axis[b][k] = float(ax[b][k])
        for k in range(4):
            m.local_rotation[b][k] = float(cm._local_rotation[b][k])
    paths = cm.fk_paths(MAX_FK_PATHS, MAX_FK_DEPTH)
    for p in range(MAX_FK_PATHS):
        for d in range(MAX_FK_DEPTH):
            m.fk_paths[p][d] = int(paths[p, d])
    return m
