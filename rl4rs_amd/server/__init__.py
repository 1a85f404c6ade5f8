"""Wire-format adapter of the reference's HTTP env server: same routes and JSON shapes as
rl4rs/server/gymHttpServer.py:239-420 and the same client-side ``HttpEnv`` (rl4rs/server/httpEnv.py:9-44), in front of the
device env.  Thin on purpose: routing and (de)serialisation only; every transition is the library's."""
from .gym_http_server import create_app, Envs           # noqa: F401
from .http_env import HttpEnv, Client                    # noqa: F401
