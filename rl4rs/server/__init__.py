from rl4rs_amd.server import *  # noqa: F401,F403
