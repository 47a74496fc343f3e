"""Heightfield grid assembly on the host (init time) — mirror of ``SubTerrain``
(``PARC/util/terrain_util.py:18-260``), ``geom_util.get_xy_points_cone:251`` and
``DeepMimicEnv.build_terrain_square / load_motion_terrain_file / load_terrain`` (``dm_env.py:105-316,447-463``).

The per-step lookups (nearest cell, ray fan) run in the HIP kernel; nothing here is on the step path.
The blocky triangle mesh the reference builds for PhysX is not produced: the dynamics kernel collides
against the cell columns of ``hf`` directly.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from parc_amd import ms_file

F32 = np.float32


class SubTerrain:
    def __init__(self, x_dim, y_dim, dx, dy, min_x, min_y):
        self.hf = np.zeros((x_dim, y_dim), F32)
        self.dims = np.array([x_dim, y_dim], np.int64)
        self.min_point = np.array([min_x, min_y], F32)
        self.dxdy = np.array([dx, dy], F32)
        self.hf_maxmin = np.zeros((x_dim, y_dim, 2), F32)
        self.hf_maxmin[..., 0] = 1.0
        self.hf_maxmin[..., 1] = -1.0

    @classmethod
    def from_ms_terrain_data(cls, td: ms_file.MSTerrainData):  # terrain_util.py:33-56
        t = cls(td.hf.shape[0], td.hf.shape[1], td.dx, td.dx, td.min_point[0], td.min_point[1])
        t.hf = np.array(td.hf, F32)
        t.hf_maxmin = np.array(td.hf_maxmin, F32)
        return t

    def to_ms_terrain_data(self):
        return ms_file.MSTerrainData(hf=self.hf, hf_maxmin=self.hf_maxmin, min_point=self.min_point, dx=float(self.dxdy[0]))

    def copy(self):
        t = SubTerrain(int(self.dims[0]), int(self.dims[1]), self.dxdy[0], self.dxdy[1], self.min_point[0], self.min_point[1])
        t.hf = self.hf.copy(); t.hf_maxmin = self.hf_maxmin.copy()
        return t

    def pad(self, padding_size: int, height=0.0):  # terrain_util.py:236-254
        p = padding_size
        self.hf = np.pad(self.hf, p, constant_values=F32(height))
        mx = self.hf_maxmin[..., 0].max(); mn = self.hf_maxmin[..., 1].min()
        self.hf_maxmin = np.stack([np.pad(self.hf_maxmin[..., 0], p, constant_values=mx),
                                   np.pad(self.hf_maxmin[..., 1], p, constant_values=mn)], axis=-1)
        self.min_point = (self.min_point - self.dxdy * F32(p)).astype(F32)
        self.dims = self.dims + 2 * p

    def get_grid_index(self, point):  # terrain_util.py:146-152 (host utility; np.rint == torch.round)
        idx = np.rint((np.asarray(point, F32) - self.min_point) / self.dxdy).astype(np.int64)
        return np.clip(idx, 0, self.dims - 1)

    def get_hf_val_from_points(self, xy):
        idx = self.get_grid_index(xy)
        return self.hf[idx[..., 0], idx[..., 1]]


def slice_terrain_around_motion(frames_xyz, terrain: "SubTerrain", padding=1.0):
    """terrain_util.py:1587-1642 (localize=True): the part of ``terrain`` under a recorded root trajectory plus ``padding``,
    shifted so that the first frame is at xy = (0, 0) and stands at height hf = 0.  Returns (SubTerrain, localized xyz).
    Same arithmetic as the torch code: fp32 tensors, ``torch.arange`` evaluated in double and stored as fp32."""
    fr = np.asarray(frames_xyz, F32)
    mn = np.array([fr[:, 0].min(), fr[:, 1].min()], F32) - F32(padding)
    mx = np.array([fr[:, 0].max(), fr[:, 1].max()], F32) + F32(padding)
    to_grid = lambda p: (np.rint((p - terrain.min_point) / terrain.dxdy) * terrain.dxdy + terrain.min_point).astype(F32)
    g0, g1 = to_grid(mn), to_grid(mx)

    def arange(start, end, step):  # torch.arange(float32 scalars): size and values computed in double
        n = int(np.ceil((float(end) - float(start)) / float(step)))
        return (float(start) + float(step) * np.arange(n, dtype=np.float64)).astype(F32)

    xs = arange(g0[0], F32(g1[0] + terrain.dxdy[0]), terrain.dxdy[0])
    ys = arange(g0[1], F32(g1[1] + terrain.dxdy[1]), terrain.dxdy[1])
    pts = np.stack(np.meshgrid(xs, ys, indexing="ij"), axis=-1)
    idx = terrain.get_grid_index(pts)
    hf = terrain.hf[idx[..., 0], idx[..., 1]].copy()
    hf_maxmin = terrain.hf_maxmin[idx[..., 0], idx[..., 1]].copy()
    canon_xy = fr[0, 0:2].copy()
    local = fr.copy()
    local[:, 0:2] -= canon_xy
    out = SubTerrain(hf.shape[0], hf.shape[1], terrain.dxdy[0], terrain.dxdy[1],
                     float(g0[0]) - float(canon_xy[0]), float(g0[1]) - float(canon_xy[1]))
    out.hf = hf
    out.hf_maxmin = hf_maxmin
    canon_z = out.get_hf_val_from_points(local[0, 0:2])
    local[:, 2] -= canon_z
    out.hf = (hf - canon_z).astype(F32)
    return out, local


def get_xy_points_cone(dx, num_neg, num_pos, num_rays_neg, num_rays_pos, angle_between_rays):
    """geom_util.py:251-272 in fp32 (torch.linspace's two-sided formula, then rotate_2d_vec per ray)."""
    dim = num_neg + num_pos + 1
    start, end = F32(-dx * num_neg), F32(dx * num_pos)
    step = F32((end - start) / F32(dim - 1))
    i = np.arange(dim)
    x = np.where(i < dim // 2, start + step * i.astype(F32), end - step * (dim - i - 1).astype(F32)).astype(F32)
    rays = []
    for r in range(num_rays_neg + 1 + num_rays_pos):
        ang = F32(-angle_between_rays * (num_rays_neg - r))
        c, s = np.cos(ang, dtype=F32), np.sin(ang, dtype=F32)
        rays.append(np.stack([x * c - F32(0.0) * s, x * s + F32(0.0) * c], axis=-1).astype(F32))
    return np.ascontiguousarray(np.concatenate(rays, axis=0), F32)


class TerrainGrid:
    """Result of a terrain build: the global ``hf`` and per-(motion, terrain) xy offsets."""

    def __init__(self, terrain: SubTerrain, motion_offsets: np.ndarray, terrains_per_motion: int):
        self.terrain = terrain
        self.motion_offsets = np.ascontiguousarray(motion_offsets, F32)  # [M][T][2]
        self.terrains_per_motion = int(terrains_per_motion)


def build_terrain_square(terrains: List[SubTerrain], horizontal_scale: float, padding: float,
                         x_offset: float = 0.0, y_offset: float = 0.0) -> TerrainGrid:
    """dm_env.py:157-316.  Tiles = ceil(sqrt(M))^2, tile = max clip dims + 2 pad cells, grid centred on the
    origin, pad cells filled with the clip's min height.  Offsets are formed in double and stored as fp32,
    which is what the reference does under the numpy 1.x it was written for (``float - np.float32``)."""
    dx = float(horizontal_scale)
    dy = dx
    num_padding_cells = padding / dx
    assert (round(num_padding_cells) - num_padding_cells) < 1e-5
    num_padding_cells = int(num_padding_cells)
    M = len(terrains)
    offsets = np.zeros((M, 1, 2), F32)
    nx = int(np.ceil(np.sqrt(M)))
    ny = nx
    dim_x = max(int(t.dims[0]) for t in terrains) + 2 * num_padding_cells
    dim_y = max(int(t.dims[1]) for t in terrains) + 2 * num_padding_cells
    first_x = -dim_x * nx * dx / 2.0
    first_y = -dim_y * ny * dy / 2.0
    og_x, og_y = x_offset, y_offset
    x_off = x_offset + first_x
    padded = []
    m = 0
    for i in range(nx):
        y_off = og_y + first_y
        for j in range(ny):
            if m > M - 1:
                break
            t = terrains[m].copy()
            t.pad(num_padding_cells, float(t.hf.min()))
            offsets[m, 0, 0] = F32(x_off - float(t.min_point[0]))
            offsets[m, 0, 1] = F32(y_off - float(t.min_point[1]))
            padded.append(t)
            m += 1
            y_off += dy * dim_y
        x_off += dx * dim_x
    g = SubTerrain(dim_x * nx, dim_y * ny, dx, dx, og_x + first_x, og_y + first_y)
    m = 0
    for i in range(nx):
        for j in range(ny):
            if m > M - 1:
                break
            t = padded[m]
            sx, sy = dim_x * i, dim_y * j
            g.hf[sx:sx + t.dims[0], sy:sy + t.dims[1]] = t.hf
            m += 1
    return TerrainGrid(g, offsets, 1)


def build_terrain_wide(terrains: List[SubTerrain], horizontal_scale: float, padding: float, terrains_per_motion: int,
                       x_offset: float = 0.0, y_offset: float = 0.0) -> TerrainGrid:
    """dm_env.py:318-445 (``terrain_build_mode: wide``): motion i occupies a strip along x, its ``terrains_per_motion``
    copies are stacked along y, ``padding`` on every side; cells between tiles stay at height 0.  Offsets are formed in
    double and stored as fp32 (numpy 1.x semantics, as in build_terrain_square)."""
    dx = float(horizontal_scale)
    dy = dx
    num_padding_cells = padding / dx
    assert (round(num_padding_cells) - num_padding_cells) < 1e-5
    num_padding_cells = int(num_padding_cells)
    M, Tn = len(terrains), int(terrains_per_motion)
    offsets = np.zeros((M, Tn, 2), F32)
    gmin = np.zeros(2, F32)
    gmax = np.zeros(2, F32)
    og_y = y_offset
    x_off = x_offset
    for i, t in enumerate(terrains):
        y_off = og_y
        for j in range(Tn):
            offsets[i, j, 0] = F32(x_off - float(t.min_point[0]))
            offsets[i, j, 1] = F32(y_off - float(t.min_point[1]))
            gmin[0] = min(float(gmin[0]), float(x_off)); gmin[1] = min(float(gmin[1]), float(y_off))
            gmax[0] = max(float(gmax[0]), float(x_off + float(t.dims[0]) * dx))
            gmax[1] = max(float(gmax[1]), float(y_off + float(t.dims[1]) * dx))
            y_off += dy * int(t.dims[1]) + padding * 2.0
        x_off += dx * int(t.dims[0]) + padding * 2.0
    dims = np.rint((gmax - gmin) / F32(dx)).astype(np.int32)
    g = SubTerrain(int(dims[0]), int(dims[1]), dx, dx, float(gmin[0]), float(gmin[1]))
    sx = 0
    for t in terrains:
        sy = 0
        for j in range(Tn):
            g.hf[sx:sx + t.dims[0], sy:sy + t.dims[1]] = t.hf
            sy += int(t.dims[1]) + 2 * num_padding_cells
        sx += int(t.dims[0]) + 2 * num_padding_cells
    return TerrainGrid(g, offsets, Tn)


def terrain_from_file(terrain: SubTerrain, num_envs: int) -> TerrainGrid:
    """dm_env.py:105-155 (``terrain_build_mode: file``): the first clip's own terrain, zero offsets."""
    return TerrainGrid(terrain.copy(), np.zeros((max(num_envs, 1), 1, 2), F32), 1)


def save_terrain(grid: TerrainGrid, path: str) -> None:
    """Reference-compatible cache (dm_env.py:431-445); mesh lists are empty (no PhysX mesh is built)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    data = ms_file.MSFileData(motion_data=None, terrain_data=grid.terrain.to_ms_terrain_data(),
                              misc_data={"terrains_per_motion": grid.terrains_per_motion,
                                         "motion_offsets": grid.motion_offsets,
                                         "all_terrain_verts": [], "all_terrain_tris": []})
    ms_file.save_ms_file(data, path)


def load_terrain(path: str) -> Optional[TerrainGrid]:
    """dm_env.py:447-463."""
    d = ms_file.load_ms_file(path)
    if d.terrain_data is None or d.misc_data is None:
        return None
    t = SubTerrain.from_ms_terrain_data(d.terrain_data)
    return TerrainGrid(t, np.asarray(d.misc_data["motion_offsets"], F32), int(d.misc_data["terrains_per_motion"]))
