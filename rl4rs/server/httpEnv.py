from rl4rs_amd.server.http_env import HttpEnv  # noqa: F401
