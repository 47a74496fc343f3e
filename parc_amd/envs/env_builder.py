"""``build_env`` (mirror of ``PARC/motion_tracker/envs/env_builder.py:8-21``).

``env_name: hip_parkour`` selects the MI355X-native env; the reference's ``ig_parkour`` name is accepted as
an alias so that its YAML files run unchanged.
"""
from parc_amd.util import path_loader


def build_env(env_file, num_envs, device, visualize, **kwargs):
    env_config = path_loader.load_config(path_loader.resolve_path(env_file))
    env_name = env_config["env_name"]
    print("Building {} env".format(env_name))
    if env_name in ("hip_parkour", "ig_parkour"):
        from parc_amd.envs import hip_parkour_env
        return hip_parkour_env.HipParkourEnv(config=env_config, num_envs=num_envs, device=device,
                                             visualize=visualize, **kwargs)
    assert False, "Unsupported env: {}".format(env_name)
