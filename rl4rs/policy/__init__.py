from rl4rs_amd.policy import *  # noqa: F401,F403
