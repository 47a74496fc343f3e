"""ctypes binding of the CPU build of the dynamics core (oracle/dyn_oracle.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdyn_oracle.so")
f32p = C.POINTER(C.c_float)


def build(force=False):
    deps = [os.path.join(_HERE, "dyn_oracle.cpp"), os.path.join(_HERE, "..", "parc_amd", "csrc", "parc_dynamics.hpp"),
            os.path.join(_HERE, "..", "include", "parc_env.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_LIB) for d in deps):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libdyn_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def _p(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(f32p)


class DynOracle:
    def __init__(self, cfg_struct):
        """cfg_struct: parc_amd.lib.ParcEnvConfig (model + dynamics + action bounds)."""
        self.lib = C.CDLL(build())
        self.lib.orc_dyn_create.restype = C.c_void_p
        self.B = cfg_struct.model.num_bodies
        self.D = cfg_struct.model.dof_size
        lo = np.array(list(cfg_struct.action_low)[: self.D], np.float32)
        hi = np.array(list(cfg_struct.action_high)[: self.D], np.float32)
        self.h = C.c_void_p(self.lib.orc_dyn_create(C.byref(cfg_struct.model), C.byref(cfg_struct.dynamics), _p(lo), _p(hi)))

    def mass_properties(self):
        mass = np.zeros(self.B, np.float32); com = np.zeros((self.B, 3), np.float32); inertia = np.zeros((self.B, 6), np.float32)
        total = C.c_float(); ncol = C.c_int()
        self.lib.orc_dyn_get_mass(self.h, _p(mass), _p(com), _p(inertia), C.byref(total), C.byref(ncol))
        return mass, com, inertia, total.value, ncol.value

    def inflate(self, eps):
        """Test hook: grow every collision shape by eps (a contact offset)."""
        self.lib.orc_dyn_inflate(self.h, C.c_float(eps))

    def set_num_substeps(self, n):
        """Test hook: substeps per control step (1 = the reported contact force is evaluated at the given pose)."""
        return int(self.lib.orc_dyn_set_nsub(self.h, C.c_int(n)))

    def set_manifold_period(self, n):
        """Contact discovery every n substeps (1 = every substep, the round-3 behaviour)."""
        return int(self.lib.orc_dyn_set_man_period(self.h, C.c_int(n)))

    def set_speculative_margin(self, m0, tv, mx):
        """Test hook: margin of the contact discovery = min(m0 + tv * approach speed, mx); (0, 0, 0) = only touching candidates become planes."""
        self.lib.orc_dyn_set_spec(self.h, C.c_float(m0), C.c_float(tv), C.c_float(mx))

    def truncated(self):
        """Collision points / segments / geoms that did not fit the model's fixed tables."""
        return int(self.lib.orc_dyn_truncated(self.h))

    def num_segments(self):
        return int(self.lib.orc_dyn_get_nseg(self.h))

    def set_num_segments(self, n):
        """Counterfactual for tests: 0 = collision points only (capsule end spheres, box corners)."""
        return int(self.lib.orc_dyn_set_nseg(self.h, C.c_int(n)))

    def segment_edge_point(self, hf, min_point, dxdy, a, b):
        hf = np.ascontiguousarray(hf, np.float32)
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); q = np.zeros(4, np.float32)
        ok = self.lib.orc_segment_edge_point(_p(hf), C.c_int(hf.shape[0]), C.c_int(hf.shape[1]), C.c_float(min_point[0]), C.c_float(min_point[1]),
                                             C.c_float(dxdy[0]), C.c_float(dxdy[1]), _p(a), _p(b), _p(q))
        return (q if ok else None)   # x, y, z, weight

    def set_gravity(self, g):
        self.lib.orc_dyn_set_gravity(self.h, C.c_float(g))

    def set_ext_acc(self, ax, ay):
        """Test hook: uniform horizontal acceleration (a tilted gravity vector)."""
        self.lib.orc_dyn_set_ext_acc(self.h, C.c_float(ax), C.c_float(ay))

    def get_contact(self):
        out = np.zeros(4, np.float32)
        self.lib.orc_dyn_get_contact(self.h, _p(out))
        return dict(kn=float(out[0]), dn=float(out[1]), dtang=float(out[2]), mu=float(out[3]))

    def set_contact(self, kn, dn, dtang, mu):
        self.lib.orc_dyn_set_contact(self.h, C.c_float(kn), C.c_float(dn), C.c_float(dtang), C.c_float(mu))

    def scale_gains(self, s):
        self.lib.orc_dyn_set_gains_scale(self.h, C.c_float(s))

    def step(self, hf, min_point, dxdy, st, action, env_off):
        """st: dict of float32 arrays root_pos[n,3], root_rot[n,4], root_vel, root_ang_vel, dof_pos[n,D], dof_vel, contact_force[n,B,3]."""
        hf = np.ascontiguousarray(hf, np.float32)
        n = st["root_pos"].shape[0]
        self.lib.orc_dyn_step(self.h, _p(hf), C.c_int(hf.shape[0]), C.c_int(hf.shape[1]), C.c_float(min_point[0]), C.c_float(min_point[1]),
                              C.c_float(dxdy[0]), C.c_float(dxdy[1]), C.c_int(n), _p(st["root_pos"]), _p(st["root_rot"]), _p(st["root_vel"]),
                              _p(st["root_ang_vel"]), _p(st["dof_pos"]), _p(st["dof_vel"]), _p(st["contact_force"]),
                              _p(np.ascontiguousarray(action, np.float32)), _p(np.ascontiguousarray(env_off, np.float32)))
