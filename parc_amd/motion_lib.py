"""Host-side motion clip loading (mirror of ``MotionLib._load_motion_file`` / ``_fetch_motion_files``,
``PARC/anim/motion_lib.py:255-423``).

Only file handling lives here.  The derived tables (root / angular / dof velocities, frame records) and
``calc_motion_frame`` run on the GPU (``parc_env_load_motions`` / ``parc_calc_motion_frame``).
"""
from __future__ import annotations

import enum
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from parc_amd import ms_file
from parc_amd.util import path_loader


class LoopMode(enum.Enum):  # motion_lib.py:14-16
    CLAMP = 0
    WRAP = 1


@dataclass
class Clip:
    name: str
    file: str
    root_pos: np.ndarray      # [n,3]
    root_rot: np.ndarray      # [n,4]
    joint_rot: np.ndarray     # [n,J,4]
    contacts: Optional[np.ndarray]  # [n,B] or None
    fps: int
    loop_mode: int
    terrain: Optional[ms_file.MSTerrainData]
    weight: float = 1.0

    @property
    def num_frames(self):
        return int(self.root_pos.shape[0])


def fetch_motion_files(motion_file):
    """motion_lib.py:403-423: a ``.yaml`` lists ``{file, weight}`` entries, anything else is one clip."""
    motion_file = str(path_loader.resolve_path(motion_file))
    if os.path.splitext(motion_file)[1] == ".yaml":
        cfg = path_loader.load_config(motion_file)
        files, weights = [], []
        for entry in cfg["motions"]:
            w = entry["weight"]
            assert w >= 0
            files.append(str(path_loader.resolve_path(entry["file"])))
            weights.append(float(w))
        return files, weights
    return [motion_file], [1.0]


_DECODED = {}  # realpath -> decoded file (datasets list the same file under many names / symlinks)


def load_clip(path: str, weight: float = 1.0) -> Clip:
    real = os.path.realpath(path)
    d = _DECODED.get(real)
    if d is None:
        d = _DECODED[real] = ms_file.load_ms_file(path, load_misc=False)
    m = d.motion_data
    if m is None:
        raise ValueError(f"{path}: no motion_data")
    n = m.root_pos.shape[0]
    assert n == m.root_rot.shape[0] == m.joint_rot.shape[0]
    return Clip(name=os.path.basename(os.path.splitext(path)[0]), file=path,
                root_pos=np.ascontiguousarray(m.root_pos, np.float32),
                root_rot=np.ascontiguousarray(m.root_rot, np.float32),
                joint_rot=np.ascontiguousarray(m.joint_rot, np.float32),
                contacts=None if m.body_contacts is None else np.ascontiguousarray(m.body_contacts, np.float32),
                fps=int(m.fps), loop_mode=LoopMode[m.loop_mode].value, terrain=d.terrain_data, weight=weight)


def load_motion_file(motion_file, verbose=True) -> List[Clip]:
    mf = str(path_loader.resolve_path(motion_file))
    if os.path.splitext(mf)[1] == ".yaml":
        spec = path_loader.load_config(mf).get("synthetic")
        if spec is not None:  # benchmark libraries generated in memory (parc_amd/util/synth_dataset.py)
            from parc_amd.util import synth_dataset
            base = load_motion_file(spec["base"], verbose=False)
            yaw = bool(spec.get("yaw", True))
            if verbose:
                print("Synthetic library: {:d} clips from {:d} ({:s})".format(int(spec["count"]), len(base), "yaw-rotated" if yaw else "replicated"))
            return synth_dataset.make_library(base, int(spec["count"]), yaw=yaw, pad=float(spec.get("pad", 3.2)), weight_by_length=yaw)
    files, weights = fetch_motion_files(motion_file)
    clips, names = [], set()
    for i, (f, w) in enumerate(zip(files, weights)):
        if verbose and (len(files) < 1000 or i % 500 == 0):
            print("Loading {:d}/{:d} motion files: {:s}".format(i + 1, len(files), f))
        c = load_clip(f, w)
        assert c.name not in names, c.name + " is a repeat. full path: " + f  # motion_lib.py:299
        names.add(c.name)
        clips.append(c)
    return clips


def pack_clips(clips: List[Clip], num_bodies: int):
    """Concatenate clips for ``ParcMotionClips`` (contacts default to zeros, motion_lib.py:345-347)."""
    nf = np.array([c.num_frames for c in clips], np.int32)
    fps = np.array([c.fps for c in clips], np.int32)
    lm = np.array([c.loop_mode for c in clips], np.int32)
    w = np.array([c.weight for c in clips], np.float64)
    rp = np.ascontiguousarray(np.concatenate([c.root_pos for c in clips]), np.float32)
    rr = np.ascontiguousarray(np.concatenate([c.root_rot for c in clips]), np.float32)
    jr = np.ascontiguousarray(np.concatenate([c.joint_rot for c in clips]), np.float32)
    ct = np.ascontiguousarray(np.concatenate([
        c.contacts if c.contacts is not None else np.zeros((c.num_frames, num_bodies), np.float32) for c in clips]), np.float32)
    return dict(num_frames=nf, fps=fps, loop_modes=lm, weights=w, root_pos=rp, root_rot=rr, joint_rot=jr, contacts=ct)
