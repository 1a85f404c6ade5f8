"""Drop-in alias: ``import rl4rs`` resolves the reference's import paths to the MI355X implementation
in ``rl4rs_amd`` (``from rl4rs.env.slate import SlateRecEnv, SlateState`` etc. keep working)."""
from rl4rs_amd import make, ENV_IDS, __version__  # noqa: F401
