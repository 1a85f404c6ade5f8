from rl4rs_amd.nets.dien import *  # noqa: F401,F403
