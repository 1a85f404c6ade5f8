from rl4rs_amd.env import RecDataBase, RecSimBase, RecEnvBase, RecState

__all__ = ["RecDataBase", "RecSimBase", "RecEnvBase", "RecState"]
