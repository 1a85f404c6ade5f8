from rl4rs_amd.utils.rllib_vector_env import MyVectorEnvWrapper  # noqa: F401
