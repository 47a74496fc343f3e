"""Synthetic motion-terrain libraries for the benchmark configurations of SURVEY.md section 8(d).

The iter-0 PARC dataset (about 16 000 motion-terrain pairs, ``doc/parc_guide.md:59``) is not in the container and cannot
be fetched, so cfg 3 / 4 / 5 run on libraries generated from the bundled clips:

* cfg 3: the bundled clips replicated to M entries (names suffixed, content unchanged);
* cfg 4 / 5: M pseudo-clips, clip ``i`` = bundled clip ``i mod n`` with its root yaw rotated by ``2 pi i / M`` and
  sampling weight = clip length in seconds.  The clip's terrain is rotated with it (nearest-cell resampling onto a new
  axis-aligned grid of the same spacing) and cropped to the motion's footprint + ``pad`` metres, so the character still
  walks on its ground and the global grid of 16 384 tiles stays a few hundred MB.

Clips are produced in memory (``make_library``); ``write_library`` stores the same clips in the reference's
motion-terrain container (``file_io.py:87-109`` via ``ms_file.save_ms_file``) plus the ``motions:`` YAML that
``MotionLib`` / ``motion_lib.load_motion_file`` read.  A motion YAML may also say::

    synthetic: {base: <motions.yaml or one .pkl>, count: 16384, yaw: true, pad: 3.2}

which ``motion_lib.load_motion_file`` expands through ``make_library`` without touching the disk.
"""
from __future__ import annotations

import os
from typing import List

import numpy as np
import yaml

from parc_amd import ms_file

F32 = np.float32


def _quat_mul(a, b):  # xyzw, broadcasting
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def yaw_rotate_clip(clip, yaw: float, name: str, pad: float = 3.2):
    """``clip`` (a ``motion_lib.Clip``) rotated by ``yaw`` about the z axis through the origin of its coordinates."""
    from parc_amd.motion_lib import Clip
    c, s = np.cos(yaw), np.sin(yaw)
    rp = clip.root_pos.astype(np.float64)
    rp2 = rp.copy()
    rp2[:, 0] = c * rp[:, 0] - s * rp[:, 1]
    rp2[:, 1] = s * rp[:, 0] + c * rp[:, 1]
    qz = np.array([0.0, 0.0, np.sin(0.5 * yaw), np.cos(0.5 * yaw)])
    rr = _quat_mul(qz[None, :], clip.root_rot.astype(np.float64))
    rr /= np.linalg.norm(rr, axis=-1, keepdims=True)
    td = clip.terrain
    dx = float(td.dx)
    src_min = np.asarray(td.min_point, np.float64)
    X, Y = td.hf.shape
    # new axis-aligned grid around the rotated trajectory, snapped to multiples of dx
    lo = np.floor((rp2[:, :2].min(axis=0) - pad) / dx) * dx
    hi = np.ceil((rp2[:, :2].max(axis=0) + pad) / dx) * dx
    nx, ny = int(round((hi[0] - lo[0]) / dx)) + 1, int(round((hi[1] - lo[1]) / dx)) + 1
    gx = lo[0] + dx * np.arange(nx)
    gy = lo[1] + dx * np.arange(ny)
    px, py = np.meshgrid(gx, gy, indexing="ij")
    sx = c * px + s * py          # inverse rotation of the new cell centres
    sy = -s * px + c * py
    ix = np.clip(np.rint((sx - src_min[0]) / dx).astype(np.int64), 0, X - 1)
    iy = np.clip(np.rint((sy - src_min[1]) / dx).astype(np.int64), 0, Y - 1)
    hf = np.ascontiguousarray(td.hf[ix, iy], F32)
    hf_maxmin = np.ascontiguousarray(np.asarray(td.hf_maxmin, F32)[ix, iy])
    td2 = ms_file.MSTerrainData(hf=hf, hf_maxmin=hf_maxmin, min_point=np.array([lo[0], lo[1]], F32), dx=dx)
    return Clip(name=name, file="<synthetic>", root_pos=np.ascontiguousarray(rp2, F32), root_rot=np.ascontiguousarray(rr, F32),
                joint_rot=clip.joint_rot, contacts=clip.contacts, fps=clip.fps, loop_mode=clip.loop_mode, terrain=td2,
                weight=clip.weight)


def make_library(base_clips: List, count: int, yaw: bool = True, pad: float = 3.2, weight_by_length: bool = True):
    """``count`` clips from ``base_clips``: entry i = base clip i mod n, yaw 2 pi i / count."""
    from dataclasses import replace
    n = len(base_clips)
    out = []
    for i in range(count):
        b = base_clips[i % n]
        name = "%s_s%05d" % (b.name, i)
        if yaw:
            cl = yaw_rotate_clip(b, 2.0 * np.pi * i / count, name, pad)
        else:
            cl = replace(b, name=name)
        if weight_by_length:
            cl.weight = float((cl.num_frames - 1) / cl.fps)
        out.append(cl)
    return out


def write_library(clips: List, out_dir: str) -> str:
    """Store ``clips`` as motion-terrain files + a ``motions:`` YAML; returns the YAML path."""
    os.makedirs(out_dir, exist_ok=True)
    ents = []
    for cl in clips:
        md = ms_file.MSMotionData(root_pos=cl.root_pos, root_rot=cl.root_rot, joint_rot=cl.joint_rot, body_contacts=cl.contacts,
                                  fps=int(cl.fps), loop_mode="CLAMP" if cl.loop_mode == 0 else "WRAP")
        path = os.path.join(out_dir, cl.name + ".pkl")
        ms_file.save_ms_file(ms_file.MSFileData(motion_data=md, terrain_data=cl.terrain, misc_data=None), path)
        ents.append({"file": path, "weight": float(cl.weight)})
    ypath = os.path.join(out_dir, "motions.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump({"motions": ents}, f)
    return ypath


def write_spec(path: str, base: str, count: int, yaw: bool = True, pad: float = 3.2) -> str:
    """A motion YAML that names the generator instead of files (expanded in memory by ``load_motion_file``)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        yaml.safe_dump({"synthetic": {"base": base, "count": int(count), "yaw": bool(yaw), "pad": float(pad)}}, f)
    return path


if __name__ == "__main__":  # python -m parc_amd.util.synth_dataset <base motions.yaml> <count> <out_dir> [--no-yaw]
    import sys
    from parc_amd import motion_lib
    base = motion_lib.load_motion_file(sys.argv[1], verbose=False)
    lib = make_library(base, int(sys.argv[2]), yaw="--no-yaw" not in sys.argv)
    print(write_library(lib, sys.argv[3]))
