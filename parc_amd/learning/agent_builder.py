"""``build_agent`` (mirror of ``PARC/motion_tracker/learning/agent_builder.py:9-24``)."""
from parc_amd.util import path_loader
from parc_amd.util.logger import Logger


def build_agent(agent_file, env, device):
    from parc_amd.learning import dm_ppo_agent
    agent_config = path_loader.load_config(path_loader.resolve_path(agent_file))
    agent_name = agent_config["agent_name"]
    Logger.print("Building {} agent".format(agent_name))
    if agent_name in (dm_ppo_agent.DMPPOAgent.NAME, "PPO"):
        agent = dm_ppo_agent.DMPPOAgent(config=agent_config, env=env, device=device)
    else:
        raise AssertionError("Unsupported agent: {}".format(agent_name))
    Logger.print("Total parameter count: {}".format(agent.calc_num_params()))
    return agent
