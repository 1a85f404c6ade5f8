from .base import RecDataBase, RecSimBase, RecEnvBase, RecState

__all__ = ["RecDataBase", "RecSimBase", "RecEnvBase", "RecState"]
