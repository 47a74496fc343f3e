"""MJCF character → flat joint/body tables for the HIP kernels.

Host-side mirror of ``PARC/anim/kin_char_model.py`` (``load_char_file:235``,
``_parse_joint:821``, ``_parse_sphere_joint:869``, ``_label_dof_indices:936``): a DFS over
``<worldbody><body>`` produces ``body_names``, ``parent_indices``, ``local_translation``,
``local_rotation`` (xyzw), joint types (3 consecutive hinges => one SPHERICAL exp-map
joint, 1 hinge => HINGE, none => FIXED) and dof offsets.  The arithmetic (dof<->quat, FK,
dof velocities) is NOT here: it runs in the HIP library (``parc_amd/csrc``); this module
only parses and packs tables, plus what the dynamics kernel needs from the same file
(geoms/densities -> masses and inertias, per-dof PD stiffness/damping/armature, motor
gears), which the reference hands to Isaac Gym by file name (``ig_char_env.py:121-136``).
"""
from __future__ import annotations

import enum
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


class JointType(enum.IntEnum):  # reference: kin_char_model.py:15-19
    ROOT = 0
    HINGE = 1
    SPHERICAL = 2
    FIXED = 3


class GeomType(enum.IntEnum):
    BOX = 0
    SPHERE = 1
    CAPSULE = 2


@dataclass
class Geom:
    name: str
    body_id: int
    shape: GeomType
    # sphere: pos, size[0]=radius; box: pos, size=half extents, quat; capsule: p0, p1, size[0]=radius
    pos: np.ndarray
    pos2: np.ndarray
    size: np.ndarray
    quat: np.ndarray  # xyzw
    density: float


@dataclass
class Joint:
    name: str
    joint_type: JointType
    axis: Optional[np.ndarray] = None  # hinge axis
    limits: Optional[np.ndarray] = None  # radians; hinge [2], spherical [3,2]
    dof_idx: int = -1
    stiffness: List[float] = field(default_factory=list)  # per dof
    damping: List[float] = field(default_factory=list)
    armature: List[float] = field(default_factory=list)
    xml_names: List[str] = field(default_factory=list)

    def get_dof_dim(self) -> int:
        return {JointType.ROOT: 0, JointType.HINGE: 1, JointType.SPHERICAL: 3, JointType.FIXED: 0}[self.joint_type]


def _floats(s, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    return np.array([float(v) for v in s.split()], dtype=np.float64)


class CharModel:
    """Tables parsed from an MJCF file.  Attribute names follow the reference class."""

    def __init__(self, char_file: Optional[str] = None):
        self._body_names: List[str] = []
        self._parent_indices = np.zeros(0, np.int32)
        self._local_translation = np.zeros((0, 3), np.float32)
        self._local_rotation = np.zeros((0, 4), np.float32)
        self._joints: List[Joint] = []
        self._geoms: List[List[Geom]] = []
        self._contact_body_names: List[str] = []
        self._dof_size = 0
        if char_file is not None:
            self.load_char_file(char_file)

    # ------------------------------------------------------------------ parsing
    def load_char_file(self, char_file: str) -> None:
        tree = ET.parse(char_file)
        root = tree.getroot()
        world = root.find("worldbody")
        assert world is not None
        body_root = world.find("body")
        assert body_root is not None

        default_joint_type, jdef, gdef = self._parse_defaults(root)
        self._joint_defaults = jdef
        self._geom_defaults = gdef

        names, parents, trans, rots, joints, geoms = [], [], [], [], [], []

        def add_body(node, parent_index, body_index):
            name = node.attrib.get("name")
            assert name is not None, "Body element is missing required 'name' attribute in character file."
            pos = _floats(node.attrib.get("pos"), [0.0, 0.0, 0.0])
            q = node.attrib.get("quat")
            if q is None:
                rot = np.array([0.0, 0.0, 0.0, 1.0])
            else:
                wxyz = _floats(q)
                rot = np.array([wxyz[1], wxyz[2], wxyz[3], wxyz[0]])
            if body_index == 0:
                joint = Joint(name="root", joint_type=JointType.ROOT)
            else:
                joint = self._parse_joint(name, node.findall("joint"), default_joint_type)
            names.append(name); parents.append(parent_index); trans.append(pos); rots.append(rot)
            joints.append(joint)
            geoms.append(self._parse_geoms(node, name, body_index))
            curr = body_index
            body_index += 1
            for child in node.findall("body"):
                body_index = add_body(child, curr, body_index)
            return body_index

        add_body(body_root, -1, 0)

        contact_info = root.find("custom_contact_info")
        if contact_info is not None:
            contact_names = []
            for elem in contact_info.findall("body"):
                nm = elem.get("name")
                assert nm in names
                contact_names.append(nm)
        else:
            contact_names = list(names)

        self._body_names = names
        self._parent_indices = np.array(parents, dtype=np.int32)
        self._local_translation = np.array(trans, dtype=np.float64).astype(np.float32)
        self._local_rotation = np.array(rots, dtype=np.float64).astype(np.float32)
        self._joints = joints
        self._geoms = geoms
        self._contact_body_names = contact_names
        self._contact_body_ids = np.array([names.index(n) for n in contact_names], dtype=np.int32)

        dof_idx = 0
        for j in joints:  # reference: _label_dof_indices:936
            j.dof_idx = dof_idx
            dof_idx += j.get_dof_dim()
        self._dof_size = dof_idx

        self._motor_gears = self._parse_actuators(root)

    @staticmethod
    def _parse_defaults(root):
        default_joint_type = None
        jdef = {}
        gdef = {}
        d = root.find("default")
        if d is not None:
            for sub in d.findall("default"):
                if sub.attrib.get("class") == "body":
                    jd = sub.find("joint")
                    if jd is not None:
                        default_joint_type = jd.attrib.get("type")
                        jdef = dict(jd.attrib)
                    gd = sub.find("geom")
                    if gd is not None:
                        gdef = dict(gd.attrib)
                    break
        return default_joint_type, jdef, gdef

    def _joint_attr(self, node, key, fallback):
        v = node.attrib.get(key)
        if v is None:
            v = self._joint_defaults.get(key)
        return float(v) if v is not None else fallback

    def _parse_joint(self, body_name, nodes, default_joint_type) -> Joint:
        n = len(nodes)
        if n == 0:
            return Joint(name=body_name, joint_type=JointType.FIXED)
        for nd in nodes:
            p = nd.attrib.get("pos")
            if p is not None and np.any(_floats(p)):
                raise ValueError("Joint offsets are not supported")
        if n == 3:  # reference: _parse_sphere_joint:869
            limits = []
            for nd in nodes:
                t = nd.attrib.get("type") or default_joint_type
                if t != "hinge":
                    raise ValueError("Invalid format for a spherical joint")
                rng = nd.attrib.get("range")
                if rng is None:
                    raise ValueError("Need joint limits")
                limits.append(_floats(rng))
            lim = (np.stack(limits).astype(np.float32) * np.float32(np.pi / 180.0)).astype(np.float32)
            name = nodes[0].attrib.get("name")
            name = name[: name.rfind("_")]
            return Joint(name=name, joint_type=JointType.SPHERICAL, limits=lim,
                         stiffness=[self._joint_attr(nd, "stiffness", 0.0) for nd in nodes],
                         damping=[self._joint_attr(nd, "damping", 0.0) for nd in nodes],
                         armature=[self._joint_attr(nd, "armature", 0.0) for nd in nodes],
                         xml_names=[nd.attrib.get("name") for nd in nodes])
        if n == 1:
            nd = nodes[0]
            t = nd.attrib.get("type") or default_joint_type
            if t == "fixed":
                return Joint(name=body_name, joint_type=JointType.FIXED)
            if t != "hinge":
                raise ValueError(f"Unsupported joint type: {t}")
            rng = nd.attrib.get("range")
            if rng is None:
                raise ValueError("Need joint limits")
            lim = (_floats(rng).astype(np.float32) * np.float32(np.pi / 180.0)).astype(np.float32)
            return Joint(name=nd.attrib.get("name"), joint_type=JointType.HINGE,
                         axis=_floats(nd.attrib.get("axis")).astype(np.float32), limits=lim,
                         stiffness=[self._joint_attr(nd, "stiffness", 0.0)],
                         damping=[self._joint_attr(nd, "damping", 0.0)],
                         armature=[self._joint_attr(nd, "armature", 0.0)],
                         xml_names=[nd.attrib.get("name")])
        raise ValueError("Series joints are not supported.")

    def _parse_geoms(self, node, body_name, body_id) -> List[Geom]:
        out = []
        default_type = self._geom_defaults.get("type", "sphere")
        for gi, g in enumerate(node.findall("geom")):
            t = g.attrib.get("type", default_type)
            name = g.attrib.get("name") or f"{body_name}_geom_{gi}"
            density = float(g.attrib.get("density", self._geom_defaults.get("density", 1000.0)))
            q = g.attrib.get("quat")
            if q is None:
                quat = np.array([0.0, 0.0, 0.0, 1.0])
            else:
                w = _floats(q)
                quat = np.array([w[1], w[2], w[3], w[0]])
            if t == "sphere":
                out.append(Geom(name, body_id, GeomType.SPHERE, _floats(g.attrib.get("pos"), [0, 0, 0]),
                                np.zeros(3), _floats(g.attrib.get("size"), [0.1]), quat, density))
            elif t == "box":
                out.append(Geom(name, body_id, GeomType.BOX, _floats(g.attrib.get("pos"), [0, 0, 0]),
                                np.zeros(3), _floats(g.attrib.get("size")), quat, density))
            elif t == "capsule":
                ft = _floats(g.attrib.get("fromto"))
                out.append(Geom(name, body_id, GeomType.CAPSULE, ft[0:3], ft[3:6],
                                _floats(g.attrib.get("size")), quat, density))
            else:
                raise ValueError(f"Unsupported geom type for dynamics: {t}")
        return out

    def _parse_actuators(self, root):
        gears = {}
        act = root.find("actuator")
        if act is not None:
            for m in act.findall("motor"):
                gears[m.attrib.get("joint")] = float(m.attrib.get("gear", 1.0))
        return gears

    # ------------------------------------------------------------------ queries (reference names)
    def get_body_names(self):
        return self._body_names

    def get_num_bodies(self):
        return len(self._body_names)

    def get_num_joints(self):
        return len(self._joints)

    def get_num_non_root_joints(self):
        return len(self._joints) - 1

    def get_num_contact_bodies(self):
        return len(self._contact_body_names)

    def get_contact_body_ids(self):
        return self._contact_body_ids

    def get_dof_size(self):
        return self._dof_size

    def get_joint(self, j) -> Joint:
        assert j > 0
        return self._joints[j]

    def get_joint_dof_idx(self, j):
        return self.get_joint(j).dof_idx

    def get_joint_dof_dim(self, j):
        return self.get_joint(j).get_dof_dim()

    def get_parent_id(self, j):
        return int(self._parent_indices[j])

    def get_body_id(self, body_name):
        return self._body_names.index(body_name)

    def get_body_name(self, body_id):
        return self._body_names[body_id]

    # ------------------------------------------------------------------ packed tables for the C-ABI
    def joint_type_array(self):
        return np.array([int(j.joint_type) for j in self._joints], dtype=np.int32)

    def joint_axis_array(self):
        ax = np.zeros((self.get_num_joints(), 3), dtype=np.float32)
        for i, j in enumerate(self._joints):
            if j.axis is not None:
                ax[i] = j.axis
        return ax

    def dof_idx_array(self):
        return np.array([j.dof_idx for j in self._joints], dtype=np.int32)

    def dof_limits(self):
        """(lower, upper) per dof in radians — reference ``_gather_joint_limits:920``."""
        lo = np.zeros(self._dof_size, np.float32)
        hi = np.zeros(self._dof_size, np.float32)
        for j in self._joints:
            d = j.get_dof_dim()
            if d == 1:
                lo[j.dof_idx] = j.limits[0]; hi[j.dof_idx] = j.limits[1]
            elif d == 3:
                lo[j.dof_idx:j.dof_idx + 3] = j.limits[:, 0]; hi[j.dof_idx:j.dof_idx + 3] = j.limits[:, 1]
        return lo, hi

    def dof_pd_params(self):
        """Per-dof (stiffness, damping, armature, effort limit) from the MJCF."""
        n = self._dof_size
        kp = np.zeros(n, np.float32); kd = np.zeros(n, np.float32)
        arm = np.zeros(n, np.float32); eff = np.zeros(n, np.float32)
        for j in self._joints:
            d = j.get_dof_dim()
            for k in range(d):
                kp[j.dof_idx + k] = j.stiffness[k]
                kd[j.dof_idx + k] = j.damping[k]
                arm[j.dof_idx + k] = j.armature[k]
                eff[j.dof_idx + k] = self._motor_gears.get(j.xml_names[k], 0.0)
        return kp, kd, arm, eff

    def fk_paths(self, max_paths=8, max_depth=8):
        """Root-to-leaf body paths (root excluded), padded with -1: the FK lane map.

        One lane walks one path, so sibling chains run in parallel without cross-lane
        traffic; shared prefixes (torso) are recomputed, which is cheaper than a
        level-synchronous sweep for a depth-4 tree.
        """
        nb = self.get_num_bodies()
        children = [[] for _ in range(nb)]
        for b in range(1, nb):
            children[int(self._parent_indices[b])].append(b)
        leaves = [b for b in range(1, nb) if not children[b]]
        paths = []
        for leaf in leaves:
            p = []
            b = leaf
            while b > 0:
                p.append(b)
                b = int(self._parent_indices[b])
            p.reverse()
            paths.append(p)
        if len(paths) > max_paths or any(len(p) > max_depth for p in paths):
            raise ValueError("character tree exceeds the FK lane map (<=8 leaves, depth <=8)")
        out = -np.ones((max_paths, max_depth), dtype=np.int32)
        for i, p in enumerate(paths):
            out[i, : len(p)] = p
        return out
