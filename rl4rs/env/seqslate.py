from rl4rs_amd.env.seqslate import SeqSlateState, SeqSlateRecEnv  # noqa: F401
