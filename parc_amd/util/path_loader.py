"""``$DATA_DIR`` substitution + YAML config loading (mirror of ``PARC/util/path_loader.py:19-70``).

``DATA_DIR`` comes from, in order: the ``PARC_DATA_DIR`` environment variable, ``DATA_DIR`` in
``data/configs/user_config.yaml`` (the reference's mechanism, if that file exists and names an existing
directory), else the repository's own ``data/`` directory.  Relative paths are resolved against the
repository root so that the bundled configs work from any working directory.
"""
import os
from pathlib import Path

import yaml

DATA_DIR_KEY = "DATA_DIR"
DATA_PLACEHOLDER = "$DATA_DIR"
REPO_ROOT = Path(__file__).resolve().parents[2]
USER_CONFIG_FILENAME = "data/configs/user_config.yaml"


def _load_data_dir() -> Path:
    env = os.environ.get("PARC_DATA_DIR")
    if env:
        return Path(env).expanduser()
    user_cfg = REPO_ROOT / USER_CONFIG_FILENAME
    if user_cfg.is_file():
        cfg = yaml.safe_load(user_cfg.read_text()) or {}
        if DATA_DIR_KEY in cfg:
            p = Path(os.path.expandvars(str(cfg[DATA_DIR_KEY]))).expanduser()
            if p.is_absolute() and p.is_dir():
                return p
    return REPO_ROOT / "data"


def _apply_data_dir(value, data_dir: Path):
    if isinstance(value, dict):
        return {k: _apply_data_dir(v, data_dir) for k, v in value.items()}
    if isinstance(value, list):
        return [_apply_data_dir(v, data_dir) for v in value]
    if isinstance(value, tuple):
        return tuple(_apply_data_dir(v, data_dir) for v in value)
    if isinstance(value, Path):
        return Path(str(value).replace(DATA_PLACEHOLDER, str(data_dir)))
    if isinstance(value, str):
        return value.replace(DATA_PLACEHOLDER, str(data_dir))
    return value


def resolve_path(path_value) -> Path:
    resolved = _apply_data_dir(path_value, _load_data_dir())
    if not isinstance(resolved, (str, Path)):
        raise AssertionError(f"Path must be a string or Path, got {type(resolved)}")
    p = Path(os.path.expandvars(str(resolved))).expanduser()
    if not p.is_absolute() and not p.exists() and (REPO_ROOT / p).exists():
        p = REPO_ROOT / p
    return p


def load_config(config_path) -> dict:
    path = resolve_path(config_path)
    if not path.is_file():
        raise AssertionError(f"Config file does not exist: {path}")
    config = yaml.safe_load(path.read_text())
    return _apply_data_dir(config, _load_data_dir())
