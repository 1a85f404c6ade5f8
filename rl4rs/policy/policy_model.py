from rl4rs_amd.policy.policy_model import *  # noqa: F401,F403
from rl4rs_amd.policy.policy_model import policy_model  # noqa: F401
