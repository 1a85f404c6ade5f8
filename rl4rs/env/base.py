from rl4rs_amd.env.base import *  # noqa: F401,F403
from rl4rs_amd.env.base import single_elem_support, RecState, RecDataBase, RecSimBase, RecEnvBase  # noqa: F401
