from rl4rs_amd.env.slate import SlateState, SlateRecEnv  # noqa: F401
