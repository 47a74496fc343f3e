"""Motion-terrain ``.pkl`` container: data-only reader and reference-compatible writer.

Mirrors the on-disk format of the reference (``PARC/util/file_io.py:1-44,65-137``):
the outer pickle is a dict ``{"motion_data": bytes|None, "terrain_data": bytes|None,
"misc_data": bytes|None}`` and each value is an independent pickle of a plain dict.

Reading never goes through :mod:`pickle`.  The payloads are tokenised with
:func:`pickletools.genops` (a disassembler: it parses opcodes and runs nothing) and
interpreted by the tiny stack machine below, which only knows how to build
``dict / list / tuple / str / bytes / int / float / bool / None`` and ``numpy.ndarray``.
Globals named by the file are never imported or called: a ``GLOBAL`` becomes an inert
:class:`_Sym` token and ``REDUCE`` / ``BUILD`` are accepted only for the exact
``numpy`` array-reconstruction pattern, which is re-implemented here from the raw
bytes.  Anything else raises :class:`UnsafePickleError` (or, with ``strict=False`` on
the ``misc_data`` payload, that payload is dropped and reported as ``None``).

Writing uses :mod:`pickle` on plain dicts of numpy arrays we own, producing files the
reference's ``file_io.load_ms_file`` (:111) reads unchanged.
"""
from __future__ import annotations

import pickle
import pickletools
from dataclasses import asdict, dataclass
from typing import Any, MutableMapping, Optional

import numpy as np

MOTION_DATA_KEY = "motion_data"
TERRAIN_DATA_KEY = "terrain_data"
MISC_DATA_KEY = "misc_data"


class UnsafePickleError(ValueError):
    """The payload needs something other than plain data + numpy arrays."""


@dataclass
class MSMotionData:  # reference: file_io.py:65-72
    root_pos: np.ndarray  # [n, 3]
    root_rot: np.ndarray  # [n, 4] quat (x, y, z, w)
    joint_rot: np.ndarray  # [n, n_joints, 4]
    body_contacts: Optional[np.ndarray]  # [n, n_body]
    fps: int
    loop_mode: str


@dataclass
class MSTerrainData:  # reference: file_io.py:74-79
    hf: np.ndarray  # [X, Y]
    hf_maxmin: np.ndarray  # [X, Y, 2]
    min_point: np.ndarray  # [2]
    dx: float


@dataclass
class MSFileData:  # reference: file_io.py:81-85
    motion_data: Optional[MSMotionData]
    terrain_data: Optional[MSTerrainData]
    misc_data: Optional[MutableMapping[str, Any]]


# --------------------------------------------------------------------------------------
# data-only decoder
# --------------------------------------------------------------------------------------
class _Sym:
    """Inert stand-in for a global named by the file (never imported)."""

    __slots__ = ("name",)

    def __init__(self, name: str):
        self.name = name

    def __repr__(self):
        return f"<sym {self.name}>"


class _Mark:
    pass


class _PendingArray:
    """Result of ``_reconstruct(ndarray, (0,), b'b')`` waiting for its BUILD state."""


class _Dtype:
    def __init__(self, code: str):
        self.code = code
        self.byteorder = "="


_RECONSTRUCT = {"numpy.core.multiarray._reconstruct", "numpy._core.multiarray._reconstruct"}
_NDARRAY = {"numpy.ndarray"}
_DTYPE = {"numpy.dtype"}
_SCALAR = {"numpy.core.multiarray.scalar", "numpy._core.multiarray.scalar"}
_ALLOWED_DTYPES = {"f2", "f4", "f8", "i1", "i2", "i4", "i8", "u1", "u2", "u4", "u8", "b1"}


def _reduce(func, args):
    if not isinstance(func, _Sym):
        raise UnsafePickleError(f"REDUCE on non-global {func!r}")
    if func.name in _RECONSTRUCT:
        if len(args) != 3 or not isinstance(args[0], _Sym) or args[0].name not in _NDARRAY:
            raise UnsafePickleError("unexpected _reconstruct arguments")
        return _PendingArray()
    if func.name in _DTYPE:
        code = args[0]
        if not isinstance(code, str) or code not in _ALLOWED_DTYPES:
            raise UnsafePickleError(f"dtype {code!r} not allowed")
        return _Dtype(code)
    if func.name in _SCALAR:
        dt, raw = args
        if not isinstance(dt, _Dtype) or not isinstance(raw, (bytes, bytearray)):
            raise UnsafePickleError("unexpected numpy scalar arguments")
        return np.frombuffer(raw, dtype=np.dtype(dt.byteorder + dt.code))[0]
    raise UnsafePickleError(f"global {func.name!r} is not plain data")


def _build(obj, state):
    if isinstance(obj, _Dtype):
        # (version, byteorder, subdescr, names, fields, itemsize, align, flags)
        if not isinstance(state, tuple) or len(state) < 5:
            raise UnsafePickleError("unexpected dtype state")
        if state[2] is not None or state[3] is not None or state[4] is not None:
            raise UnsafePickleError("structured dtypes are not plain data")
        bo = state[1]
        obj.byteorder = {"<": "<", ">": ">", "|": "|", "=": "="}.get(bo, "=")
        return obj
    if isinstance(obj, _PendingArray):
        # (version, shape, dtype, is_fortran, rawdata)
        if not isinstance(state, tuple) or len(state) != 5:
            raise UnsafePickleError("unexpected ndarray state")
        _, shape, dt, fortran, raw = state
        if not isinstance(dt, _Dtype) or not isinstance(raw, (bytes, bytearray)):
            raise UnsafePickleError("ndarray of objects is not plain data")
        if not all(isinstance(s, int) and s >= 0 for s in shape):
            raise UnsafePickleError("bad ndarray shape")
        arr = np.frombuffer(raw, dtype=np.dtype(obj_dtype_str(dt)))
        arr = arr.reshape(shape, order="F" if fortran else "C")
        return np.array(arr)  # own, writable copy
    raise UnsafePickleError(f"BUILD on {type(obj).__name__} is not plain data")


def obj_dtype_str(dt: _Dtype) -> str:
    bo = dt.byteorder if dt.byteorder in "<>|=" else "="
    return bo + dt.code


def loads_data_only(blob: bytes):
    """Decode a pickle byte string holding only plain data and numpy arrays."""
    stack = []
    memo = {}

    def pop_to_mark():
        items = []
        while True:
            if not stack:
                raise UnsafePickleError("mark underflow")
            x = stack.pop()
            if isinstance(x, _Mark):
                break
            items.append(x)
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(blob):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "STOP":
            break
        elif n == "MARK":
            stack.append(_Mark())
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n in ("MEMOIZE",):
            memo[len(memo)] = stack[-1]
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[int(arg)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[int(arg)])
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "UNICODE",
                   "BINBYTES", "SHORT_BINBYTES", "BINBYTES8", "BYTEARRAY8",
                   "BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG",
                   "BINFLOAT", "FLOAT", "SHORT_BINSTRING", "BINSTRING"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "TUPLE1":
            a = stack.pop()
            stack.append((a,))
        elif n == "TUPLE2":
            b = stack.pop(); a = stack.pop()
            stack.append((a, b))
        elif n == "TUPLE3":
            c = stack.pop(); b = stack.pop(); a = stack.pop()
            stack.append((a, b, c))
        elif n == "TUPLE":
            stack.append(tuple(pop_to_mark()))
        elif n == "LIST":
            stack.append(list(pop_to_mark()))
        elif n == "DICT":
            it = pop_to_mark()
            stack.append(dict(zip(it[0::2], it[1::2])))
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "APPENDS":
            it = pop_to_mark()
            stack[-1].extend(it)
        elif n == "SETITEM":
            v = stack.pop(); k = stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            it = pop_to_mark()
            d = stack[-1]
            if not isinstance(d, dict):
                raise UnsafePickleError("SETITEMS on non-dict")
            for k, v in zip(it[0::2], it[1::2]):
                d[k] = v
        elif n == "STACK_GLOBAL":
            name = stack.pop(); mod = stack.pop()
            stack.append(_Sym(f"{mod}.{name}"))
        elif n == "GLOBAL":
            stack.append(_Sym(str(arg).replace(" ", ".")))
        elif n == "REDUCE":
            args = stack.pop(); func = stack.pop()
            stack.append(_reduce(func, args))
        elif n == "BUILD":
            state = stack.pop(); obj = stack.pop()
            stack.append(_build(obj, state))
        else:
            raise UnsafePickleError(f"opcode {n} is not plain data")
    if len(stack) != 1:
        raise UnsafePickleError("malformed pickle: stack depth != 1 at STOP")
    return _check_plain(stack[0])


def _check_plain(x):
    if isinstance(x, (_Sym, _Mark, _PendingArray, _Dtype)):
        raise UnsafePickleError("unresolved non-data object in payload")
    if isinstance(x, dict):
        for k, v in x.items():
            _check_plain(k); _check_plain(v)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _check_plain(v)
    return x


# --------------------------------------------------------------------------------------
# container API (reference: file_io.load_ms_file:111, save_ms_file:87)
# --------------------------------------------------------------------------------------
def load_ms_file(filepath, load_misc: bool = True) -> MSFileData:
    with open(filepath, "rb") as f:
        container = loads_data_only(f.read())
    if not isinstance(container, dict):
        raise UnsafePickleError("container is not a dict")

    def inner(key, strict=True):
        blob = container.get(key)
        if blob is None:
            return None
        try:
            return loads_data_only(blob)
        except UnsafePickleError:
            if strict:
                raise
            return None

    motion = inner(MOTION_DATA_KEY)
    terrain = inner(TERRAIN_DATA_KEY)
    # misc payloads may hold torch tensors or project classes: not plain data -> dropped
    misc = inner(MISC_DATA_KEY, strict=False) if load_misc else None

    motion_data = None
    if motion is not None:
        motion_data = MSMotionData(
            root_pos=np.asarray(motion["root_pos"]),
            root_rot=np.asarray(motion["root_rot"]),
            joint_rot=np.asarray(motion["joint_rot"]),
            body_contacts=None if motion.get("body_contacts") is None else np.asarray(motion["body_contacts"]),
            fps=motion["fps"],
            loop_mode=motion["loop_mode"],
        )
    terrain_data = None
    if terrain is not None:
        terrain_data = MSTerrainData(
            hf=np.asarray(terrain["hf"]),
            hf_maxmin=np.asarray(terrain["hf_maxmin"]),
            min_point=np.asarray(terrain["min_point"]),
            dx=float(terrain["dx"]),
        )
    return MSFileData(motion_data=motion_data, terrain_data=terrain_data, misc_data=misc)


def save_ms_file(data: MSFileData, filepath) -> None:
    container = {
        MOTION_DATA_KEY: None if data.motion_data is None else pickle.dumps(asdict(data.motion_data)),
        TERRAIN_DATA_KEY: None if data.terrain_data is None else pickle.dumps(asdict(data.terrain_data)),
        MISC_DATA_KEY: None if data.misc_data is None else pickle.dumps(data.misc_data),
    }
    with open(filepath, "wb") as f:
        pickle.dump(container, f)
