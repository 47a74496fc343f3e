"""CPU tests of the re-authored dynamics core (parc_amd/csrc/parc_dynamics.hpp built for the host).

PhysX parity is unpinned (closed binary, absent): these are the invariants SURVEY.md §7 asks for instead —
mass properties, momentum in free flight, free fall, PD step response, resting contact on a cell column,
wall contact — run on the same code the HIP kernel compiles."""
import os

import numpy as np
import pytest

from conftest import DATA, golden


@pytest.fixture(scope="module")
def dyn():
    from oracle.binding_dyn import DynOracle
    from parc_amd.envs import scene
    from parc_amd.util import path_loader
    cfg = path_loader.load_config(os.path.join(DATA, "configs/tracker_config/dm_env_default.yaml"))
    sc = scene.build_scene(cfg, 4, verbose=False)
    return DynOracle(sc.cfg), sc


def make_state(n, B=15, D=28, z=5.0):
    st = dict(root_pos=np.zeros((n, 3), np.float32), root_rot=np.zeros((n, 4), np.float32), root_vel=np.zeros((n, 3), np.float32),
              root_ang_vel=np.zeros((n, 3), np.float32), dof_pos=np.zeros((n, D), np.float32), dof_vel=np.zeros((n, D), np.float32),
              contact_force=np.zeros((n, B, 3), np.float32))
    st["root_rot"][:, 3] = 1.0
    st["root_pos"][:, 2] = z
    return st


FLAT = (np.zeros((64, 64), np.float32), (-12.8, -12.8), (0.4, 0.4))


def com_of(oracle, orc_char, d, st):
    mass, com, _, total, _ = d.mass_properties()
    jr = oracle.dof_to_rot(orc_char, st["dof_pos"])
    bp, br = oracle.forward_kinematics(orc_char, st["root_pos"], st["root_rot"], jr)
    n = bp.shape[0]
    c = np.zeros((n, 3))
    for b in range(15):
        off = oracle.quat_rotate(br[:, b], np.tile(com[b], (n, 1)))
        c += mass[b] * (bp[:, b] + off)
    return c / total


def test_mass_properties(dyn):
    d, sc = dyn
    mass, com, inertia, total, ncol = d.mass_properties()
    assert 40.0 < total < 55.0, total           # densities of humanoid.xml give ~50 kg
    assert np.all(mass > 0.05) and ncol == 42
    assert np.all(inertia[:, :3] > 0)
    # symmetric model: left/right limbs have equal masses
    for r, l in [(3, 6), (4, 7), (5, 8), (9, 12), (10, 13), (11, 14)]:
        assert abs(mass[r] - mass[l]) < 1e-5


def test_free_fall_and_momentum(dyn, oracle, orc_char):
    d, sc = dyn
    n = 4
    rng = np.random.default_rng(0)
    st = make_state(n, z=50.0)
    st["dof_pos"][:] = 0.3 * rng.standard_normal((n, 28)).astype(np.float32)
    act = (st["dof_pos"] + 0.2 * rng.standard_normal((n, 28))).astype(np.float32)
    off = np.zeros((n, 3), np.float32)
    c0 = com_of(oracle, orc_char, d, st)
    steps = 30
    for _ in range(steps):
        d.step(*FLAT, st, act, off)
    c1 = com_of(oracle, orc_char, d, st)
    assert np.all(np.isfinite(st["dof_pos"]))
    # internal PD torques cannot move the centre of mass sideways (first-order integrator: O(dt) drift allowed);
    # vertically it is a free fall
    assert np.abs(c1[:, :2] - c0[:, :2]).max() < 1e-2
    nsub = steps * 4
    dt = 1.0 / 120.0
    z_expected = -9.81 * dt * dt * nsub * (nsub + 1) / 2.0   # semi-implicit Euler
    assert np.abs((c1[:, 2] - c0[:, 2]) - z_expected).max() < 5e-3
    assert np.abs(st["contact_force"]).max() == 0.0


def test_pd_step_response_zero_g(dyn, oracle, orc_char):
    d, sc = dyn
    d.set_gravity(0.0)
    try:
        n = 2
        st = make_state(n, z=50.0)
        rng = np.random.default_rng(1)
        lo, hi = sc.action_low.astype(np.float32), sc.action_high.astype(np.float32)
        tgt = (0.25 * rng.uniform(-1, 1, (n, 28)) * np.minimum(np.abs(lo), np.abs(hi))).astype(np.float32)
        tgt[:, [9, 13]] = 0.0
        tgt[:, 9] = 0.8; tgt[:, 13] = -0.8; tgt[:, 17] = 0.6; tgt[:, 24] = 0.6   # hinges inside their ranges
        off = np.zeros((n, 3), np.float32)
        hist = []
        for _ in range(60):
            d.step(*FLAT, st, tgt, off)
            hist.append(np.abs(st["dof_vel"]).max())
        assert np.all(np.isfinite(st["dof_pos"]))
        # hinge angles reach their targets; spherical joints reach the target rotation
        jr = oracle.dof_to_rot(orc_char, st["dof_pos"]); jt = oracle.dof_to_rot(orc_char, tgt)
        ang = oracle.quat_diff_angle(jr.reshape(-1, 4), jt.reshape(-1, 4))
        assert np.abs(ang).max() < 0.03, np.abs(ang).max()
        assert hist[-1] < 0.05 and max(hist) < 60.0   # settles, never explodes
    finally:
        d.set_gravity(-9.81)


def standing_state(n, z):
    st = make_state(n, z=z)
    return st


def test_resting_contact_flat_ground(dyn, oracle, orc_char):
    d, sc = dyn
    n = 2
    st = standing_state(n, 0.95)   # feet a few cm above the ground (root height of the T-pose is ~0.89)
    act = np.zeros((n, 28), np.float32)
    off = np.zeros((n, 3), np.float32)
    _, _, _, total, _ = d.mass_properties()
    fz = []
    for k in range(45):   # 1.5 s
        d.step(*FLAT, st, act, off)
        fz.append(st["contact_force"][:, :, 2].sum(axis=1))
        assert np.all(np.isfinite(st["root_pos"]))
        if np.abs(st["root_rot"][:, :2]).max() > 0.3:
            break   # a passive T-pose eventually topples; the checks below use the standing phase
    fz = np.array(fz)
    k_settled = min(len(fz) - 1, 20)
    # while standing, the ground carries the weight and the feet do not sink
    assert np.all(np.abs(fz[10:k_settled + 1].mean(axis=0) - total * 9.81) < 0.25 * total * 9.81), fz[10:k_settled + 1].mean(axis=0)
    jr = oracle.dof_to_rot(orc_char, st["dof_pos"])
    bp, _ = oracle.forward_kinematics(orc_char, st["root_pos"], st["root_rot"], jr)
    assert bp[:, :, 2].min() > -0.02
    # only the feet touch while standing
    assert np.abs(st["contact_force"][:, 1:9]).max() < 1e-3 or np.abs(st["root_rot"][:, :2]).max() > 0.3


def test_wall_blocks_motion(dyn, oracle, orc_char):
    d, sc = dyn
    hf = np.zeros((64, 64), np.float32)
    hf[40:, :] = 3.0    # a 3 m wall starting at x = -12.8 + 39.5*0.4 = 3.0
    n = 1
    st = make_state(n, z=1.5)
    st["root_vel"][:, 0] = 4.0      # thrown at the wall
    act = np.zeros((n, 28), np.float32)
    off = np.zeros((n, 3), np.float32)
    d.set_gravity(0.0)
    try:
        for _ in range(60):
            d.step(hf, (-12.8, -12.8), (0.4, 0.4), st, act, off)
        assert np.all(np.isfinite(st["root_pos"]))
        assert st["root_pos"][0, 0] < 3.0 + 0.05, st["root_pos"]   # did not tunnel through the wall face at x = 3.0
    finally:
        d.set_gravity(-9.81)


def test_step_edge_supports_foot(dyn, oracle, orc_char):
    """Standing on a raised block: the character rests at block height, not at the surrounding ground."""
    d, sc = dyn
    hf = np.zeros((64, 64), np.float32)
    hf[30:35, 30:35] = 0.8          # 2 m x 2 m block, top at 0.8 m, centred near the origin
    n = 1
    st = make_state(n, z=0.8 + 0.95)
    st["root_pos"][0, 0] = -12.8 + 32 * 0.4
    st["root_pos"][0, 1] = -12.8 + 32 * 0.4
    act = np.zeros((n, 28), np.float32)
    off = np.zeros((n, 3), np.float32)
    for _ in range(15):
        d.step(hf, (-12.8, -12.8), (0.4, 0.4), st, act, off)
    assert 0.8 + 0.8 < st["root_pos"][0, 2] < 0.8 + 1.0, st["root_pos"]


def test_body_that_crossed_a_wall_face_is_pushed_back_not_launched(dyn, oracle, orc_char):
    """Regression: a point whose centre had crossed a wall face (its cell = the wall's cell) used to be pushed out through
    the TOP of the wall, metres away: tens of kN, bodies launched at > 100 m/s.  It must leave sideways, the way it came."""
    d, sc = dyn
    hf = np.zeros((64, 64), np.float32)
    hf[40:, :] = 5.0    # 5 m wall, face at x = 3.0
    n = 3
    st = make_state(n, z=0.95)
    st["root_pos"][:, 0] = [2.55, 2.70, 2.80]   # hands / feet reach or cross the face during the fall
    st["root_vel"][:, 0] = [6.0, 3.0, 1.0]
    act = np.zeros((n, 28), np.float32)
    off = np.zeros((n, 3), np.float32)
    vmax, zmax = 0.0, 0.0
    for _ in range(90):  # 3 s: run into the wall, fall, come to rest at its foot
        d.step(hf, (-12.8, -12.8), (0.4, 0.4), st, act, off)
        assert np.all(np.isfinite(st["root_pos"]))
        vmax = max(vmax, np.linalg.norm(st["root_vel"], axis=1).max())
        zmax = max(zmax, st["root_pos"][:, 2].max())
    assert vmax < 12.0, vmax                       # nothing is launched (free-fall from 1 m is 4.4 m/s, the throw 6 m/s)
    assert zmax < 1.6, zmax                        # and nobody ends up on top of the wall
    assert st["root_pos"][:, 0].max() < 3.0 + 0.3  # the root stays on this side of the face (limbs may dent the wall by < pen_cap)
    assert np.abs(st["contact_force"]).max() < 50.0 * 9.81 * 30.0


def test_own_column_exit_rule():
    """own_column_contact through the host build: geometry only."""
    import ctypes as C
    from oracle.binding_dyn import build
    lib = C.CDLL(build())
    lib.orc_own_column_contact.restype = C.c_float
    lib.orc_own_column_contact.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    n = (C.c_float * 3)()

    def q(tops5, x, y, z, r):  # tops5: own, +x, -x, +y, -y; cell (0,0) centred at the origin, 0.4 m cells
        t = (C.c_float * 5)(*tops5)
        pen = lib.orc_own_column_contact(t, x, y, z, r, n)
        return pen, (n[0], n[1], n[2])
    # above the surface: +z, pen = r - height
    pen, nn = q([1.0, 0, 0, 0, 0], 0.05, 0.0, 1.03, 0.05)
    assert nn == (0.0, 0.0, 1.0) and abs(pen - 0.02) < 1e-6
    # 1 cm inside the -x face of a 5 m wall, free ground on that side: leaves through that face
    pen, nn = q([5.0, 5.0, 0.0, 5.0, 5.0], -0.19, 0.0, 1.0, 0.04)
    assert nn == (-1.0, 0.0, 0.0) and abs(pen - 0.05) < 1e-6
    # toe corner (r = 0) 1 mm inside a 0.4 m riser while pressing 2 mm into the lower step: sideways, not 0.4 m up
    pen, nn = q([0.4, 0.4, 0.002 + 0.1, 0.4, 0.4], -0.199, 0.0, 0.1, 0.0)
    assert nn == (-1.0, 0.0, 0.0) and abs(pen - 0.001) < 1e-6
    # deep inside a plateau (all neighbours as high): only the top is a way out
    pen, nn = q([1.0, 1.0, 1.0, 1.0, 1.0], 0.1, 0.1, 0.7, 0.05)
    assert nn == (0.0, 0.0, 1.0) and abs(pen - 0.35) < 1e-6
    # just under the top of a pillar: up (3 cm) beats sideways (15 cm)
    pen, nn = q([1.0, 0.0, 0.0, 0.0, 0.0], 0.05, 0.0, 0.97, 0.0)
    assert nn == (0.0, 0.0, 1.0) and abs(pen - 0.03) < 1e-6
