#!/usr/bin/env python3
"""Developer tool: prints the PhysX-anchored transition metrics of tests/physx_transitions.py (one-step errors of the floating base from every
recorded state of dec2024_teaser_717_1_opt_dm.pkl, the closed-loop drift over 10 control steps, and the counterfactual models)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from physx_transitions import closed_loop, load, one_step  # noqa: E402

if __name__ == "__main__":
    from conftest import golden
    from oracle.binding import Oracle
    from oracle.binding_dyn import DynOracle
    from parc_amd.envs import scene
    from parc_amd.util import path_loader
    oracle = Oracle()
    cg = golden("char_model")
    oc = oracle.make_char(cg["parent"], cg["local_translation"], cg["local_rotation"], cg["joint_type"], cg["joint_axis"], cg["dof_idx"], int(cg["dof_size"]))
    cfg = path_loader.load_config(os.path.join(REPO, "data/configs/tracker_config/dm_env_default.yaml"))
    sc = scene.build_scene(cfg, 4, verbose=False)
    R = load(oracle, oc, cg)
    fd, arm = R["fd_check"]
    nonarm = [i for i in range(28) if i not in arm]
    print("velocity source check (non-arm dofs): median |obs dof_vel - central difference of the frames| = %.3f rad/s, corr = %.3f" %
          (np.median(np.abs(R["dof_vel"][:, nonarm] - fd[:, nonarm])), np.corrcoef(R["dof_vel"][:, nonarm].ravel(), fd[:, nonarm].ravel())[0, 1]))
    for eps in (0.0, 0.01):
        d = DynOracle(sc.cfg)
        if eps: d.inflate(eps)
        r = one_step(d, oracle, R)
        q = lambda x: "med %.4f q90 %.4f max %.4f" % (np.median(x), np.quantile(x, 0.9), x.max())
        print("== one step, contact offset %.3f" % eps)
        print(" fit residual (rad):", q(r["fit_residual"]))
        for k in ("e_pos", "e_z", "e_rot", "e_vel", "e_cv", "e_ff", "e_hold"):
            print(" %-7s %s" % (k, q(r[k])))
        print(" contacts: all %.3f feet %.3f foot rate sim %.3f ref %.3f  missed persistent contacts %.4f" % (r["agree_all"], r["agree_feet"], r["foot_rate_sim"], r["foot_rate_ref"], r["false_neg"]))
        print(" sum Fz / mg: med %.2f" % np.median(r["fz"] / (50.05 * 9.81)))
    d = DynOracle(sc.cfg)
    for row in closed_loop(d, oracle, R):
        print(row)
    print("== counterfactuals, closed loop h = 10")
    for name, kw in (("no friction", dict(mu=0.0)), ("contacts 100x softer", dict(kn=5e2, dn=5.0, dtang=3e2))):
        d = DynOracle(sc.cfg)
        c = d.get_contact(); c.update(kw)
        d.set_contact(c["kn"], c["dn"], c["dtang"], c["mu"])
        print(name, closed_loop(d, oracle, R)[-1])
    for per in (1, 2, 4):
        d = DynOracle(sc.cfg); d.set_manifold_period(per)
        r1 = one_step(d, oracle, R)
        print("manifold period %d: one step e_pos med %.4f e_z med %.4f feet %.3f | closed loop h=10" % (per, np.median(r1["e_pos"]), np.median(r1["e_z"]), r1["agree_feet"]), closed_loop(d, oracle, R)[-1])
    d = DynOracle(sc.cfg); d.set_num_segments(0)
    print("points only (no segments)", closed_loop(d, oracle, R)[-1])
    r = one_step(DynOracle(sc.cfg), oracle, R)
    dv_null = np.linalg.norm(R["root_vel"][1:] - R["root_vel"][:-1], axis=1)
    print("one-step velocity: |v_sim - v_rec| med %.3f  vs null |v_t+1 - v_t| med %.3f" % (np.median(r["e_vel"]), np.median(dv_null)))
