#!/usr/bin/env python3
"""Example script for reading the motion data format — counterpart of the reference's ``scripts/read_motion_data.py:1-20``
(BASELINE cfg 1, the CPU plumbing case): prints the arrays of a motion-terrain ``.pkl``.

    python scripts/read_motion_data.py [data/motion_terrains/sfu.pkl]

The file is read with the data-only decoder ``parc_amd/ms_file.py`` (nothing in it is executed).  With ``--frame T`` the clip
is additionally sampled at time T through the library's own ``calc_motion_frame`` + ``forward_kinematics`` on the GPU
(``parc_calc_motion_frame`` / ``parc_forward_kinematics``; needs a MI355X, there is no CPU path for the kernels).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from parc_amd import ms_file  # noqa: E402

MOTION_FILE = "data/motion_terrains/sfu.pkl"


def read(path):
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(REPO, path)
    return ms_file.load_ms_file(path)


def sample_on_gpu(path, t):
    """cfg 1 through the HIP path: motion frame at time ``t`` and its FK body positions."""
    import copy
    import tempfile
    import torch
    import yaml
    import ctypes as C
    from parc_amd import lib as L
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from parc_amd.util import path_loader
    cfg = copy.deepcopy(path_loader.load_config(os.path.join(REPO, "data/configs/tracker_config/dm_env_default.yaml")))
    d = tempfile.mkdtemp(prefix="parc_read_")
    with open(os.path.join(d, "motions.yaml"), "w") as f:
        yaml.safe_dump({"motions": [{"file": os.path.abspath(path), "weight": 1.0}]}, f)
    cfg["env"]["dm"]["motion_file"] = os.path.join(d, "motions.yaml")
    cfg["env"]["dm"].pop("terrain_save_path", None)
    cfg["env"].setdefault("hip", {})["enable_dynamics"] = False
    env = HipParkourEnv(cfg, 1, "cuda:0", False)
    dev = "cuda:0"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ids = torch.zeros(1, dtype=torch.int32, device=dev); tt = torch.tensor([t], dtype=torch.float32, device=dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    o = dict(root_pos=z(1, 3), root_rot=z(1, 4), root_vel=z(1, 3), root_ang_vel=z(1, 3), joint_rot=z(1, 14, 4), dof_vel=z(1, 28), contacts=z(1, 15))
    L.check(env._lib.parc_calc_motion_frame(env._handle, ids.data_ptr(), tt.data_ptr(), 1, *[o[k].data_ptr() for k in
            ["root_pos", "root_rot", "root_vel", "root_ang_vel", "joint_rot", "dof_vel", "contacts"]], st))
    bp, br = z(1, 15, 3), z(1, 15, 4)
    L.check(env._lib.parc_forward_kinematics(env._handle, o["root_pos"].data_ptr(), o["root_rot"].data_ptr(), o["joint_rot"].data_ptr(),
                                             bp.data_ptr(), br.data_ptr(), 1, st))
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy()[0] for k, v in o.items()}
    out["body_pos"] = bp.cpu().numpy()[0]
    return out


def main(argv):
    args = [a for a in argv[1:] if not a.startswith("--")]
    path = args[0] if args else MOTION_FILE
    ms_file_data = read(path)
    print(ms_file_data.motion_data.root_pos)
    print(ms_file_data.motion_data.root_rot)
    print(ms_file_data.motion_data.joint_rot)
    print(ms_file_data.motion_data.body_contacts)
    print(ms_file_data.motion_data.fps)
    if "--frame" in argv:
        t = float(argv[argv.index("--frame") + 1])
        np.set_printoptions(precision=7, suppress=True)
        fr = sample_on_gpu(path if os.path.exists(path) else os.path.join(REPO, path), t)
        print("frame at t = %g" % t)
        for k in ["root_pos", "root_rot", "joint_rot", "contacts", "body_pos"]:
            print(k, fr[k])
    return ms_file_data


if __name__ == "__main__":
    main(sys.argv)
