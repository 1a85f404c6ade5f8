"""rl4rs_amd — the RL4RS batched env.step() hot path on MI355X (gfx950).

Entry points mirror the reference: ``SlateRecEnv`` / ``SeqSlateRecEnv`` simulators, ``SlateState`` /
``SeqSlateState`` plugins, ``RecEnvBase`` gym facade registered as ``SlateRecEnv-v0`` /
``SeqSlateRecEnv-v0`` (rl4rs/__init__.py:8-16).  ``make(id, recsim=sim)`` works with or without gym.
"""
__version__ = '0.1.0'

ENV_IDS = ('SlateRecEnv-v0', 'SeqSlateRecEnv-v0')


def make(env_id, recsim=None, **kwargs):
    """gym.make(id, recsim=sim) equivalent (both ids map to RecEnvBase, rl4rs/__init__.py:8-16)."""
    if env_id not in ENV_IDS:
        raise ValueError('unknown env id %r (have %r)' % (env_id, ENV_IDS))
    from .env.base import RecEnvBase
    return RecEnvBase(recsim)


def _register_with_gym():
    try:
        from gym.envs.registration import register
    except Exception:
        return
    for env_id in ENV_IDS:
        try:
            register(id=env_id, entry_point='rl4rs_amd.env:RecEnvBase')
        except Exception:
            pass


_register_with_gym()
