"""check_total_timesteps (mava/utils/config.py:47-81): derive num_updates / total_timesteps."""
from __future__ import annotations


def check_total_timesteps(config, n_devices: int = 1):
    ubs = config.system.update_batch_size if config.arch.architecture_name == "anakin" else 1
    nd = n_devices if config.arch.architecture_name == "anakin" else 1
    if config.system.total_timesteps is None:
        config.system.num_updates = int(config.system.num_updates)
        config.system.total_timesteps = int(nd * config.system.num_updates * config.system.rollout_length * ubs * config.arch.num_envs)
    else:
        config.system.total_timesteps = int(config.system.total_timesteps)
        config.system.num_updates = int(config.system.total_timesteps // config.system.rollout_length // ubs // config.arch.num_envs // nd)
        print(f"Changing the number of updates to {config.system.num_updates}: If you want to train for a specific number of "
              "updates, please set total_timesteps to None!")
    return config
