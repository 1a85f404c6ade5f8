from rl4rs_amd.server.http_env import Client, ServerError  # noqa: F401
