"""Motion dataset YAML for the tracker (``util/create_dataset.py:20-177``): walk folders of motion-terrain files, one motion
class per first-level folder, and write ``motions: [{file, weight}]`` with weight = clip length x a per-class factor that
gives every class the same total sampling mass (``motion_class_proportions`` are all 1.0 in the reference).

Not ported: ``compute_preprocessing_data`` (the hf-mask extras the diffusion generator's training reads) — this
repository is the tracker; files are never rewritten here, they are only read with the data-only decoder."""
from pathlib import Path
from typing import List

import yaml

from parc_amd import ms_file


def create_dataset_yaml(folder_paths: List[Path], save_path: Path, cut_some_classes_in_half: bool = False,
                        motion_classes_to_cut_in_half: List[str] = (), max_terrain_dim_x: int = 10 ** 9, max_terrain_dim_y: int = 10 ** 9,
                        min_num_frames: int = 0, verbose: bool = True):
    folder_paths = [Path(p) for p in folder_paths]
    classes = []
    for fp in folder_paths:  # create_dataset.py:35-43
        for folder in sorted(p for p in fp.iterdir() if p.is_dir()):
            if "ignore" in str(folder):
                continue
            classes.append(folder.name)
    if verbose:
        print("MOTION CLASSES:"); print(classes)
    dirs = []
    for fp in folder_paths:
        dirs.extend(p for p in fp.rglob("*") if p.is_dir() and "ignore" not in str(p))
    motions = {c: [] for c in classes}
    lengths = {c: 0.0 for c in classes}
    for d in dirs:
        files = sorted(d.glob("*.pkl"))
        if cut_some_classes_in_half and any(c in str(d) for c in motion_classes_to_cut_in_half):
            files = files[::2]
        for f in files:
            data = ms_file.load_ms_file(str(f), load_misc=False)
            n = data.motion_data.root_pos.shape[0]
            if n < min_num_frames:
                if verbose:
                    print("excluding motion with too few frames:", n, "<", min_num_frames)
                continue
            hf = data.terrain_data.hf
            if hf.shape[0] > max_terrain_dim_x or hf.shape[1] > max_terrain_dim_y:
                if verbose:
                    print("Large terrain excluded"); print(f); print(hf.shape)
                continue
            length = n / data.motion_data.fps   # create_dataset.py:106 (num_frames / fps, not (n-1)/fps)
            cls = next((c for c in classes if ("/" + c + "/") in str(f) or ("\\\\" + c + "\\\\") in str(f)), None)
            assert cls is not None, ("no motion class found in ", f)
            motions[cls].append((str(f), length))
            lengths[cls] += length
    total = sum(lengths.values())
    out = []
    for c in classes:
        assert lengths[c] > 0.0, c + " has no motion"
        factor = (1.0 / len(classes)) / (lengths[c] / total)   # intended fraction / actual fraction
        if verbose:
            print(c, "total length:", lengths[c], "weight_factor", factor)
        out.extend({"file": f, "weight": l * factor} for f, l in motions[c])
    Path(save_path).write_text(yaml.dump({"motions": out}))
    return out


def create_dataset_yaml_from_config(config):
    print("Creating datatset yaml:", config["save_path"])
    return create_dataset_yaml(folder_paths=[Path(p) for p in config["folder_paths"]], save_path=Path(config["save_path"]),
                               cut_some_classes_in_half=config.get("cut_some_classes_in_half", False),
                               motion_classes_to_cut_in_half=config.get("motion_classes_to_cut_in_half", []),
                               max_terrain_dim_x=config.get("max_terrain_dim_x", 10 ** 9), max_terrain_dim_y=config.get("max_terrain_dim_y", 10 ** 9),
                               min_num_frames=config.get("min_num_frames", 0))
