from rl4rs_amd.server.gym_http_server import create_app, Envs, InvalidUsage, main  # noqa: F401
