"""CPU: wire format of the HTTP env adapter (rl4rs/server/gymHttpServer.py:239-420, httpEnv.py:9-44):
routes, JSON shapes, error replies and the HttpEnv round trip, against a stub batched env (no GPU in this test; the GPU
leg is tests/test_gpu_facade.py::test_http_env_round_trip)."""
import numpy as np
import pytest

from rl4rs_amd.env.base import _spaces


class StubEnv(object):
    """Batched env with the shapes of SlateRecEnv-v0 in rllib-mask mode: obs = list[B] of dict(action_mask[A], obs[256])."""

    def __init__(self, config):
        self.B, self.A = config['batch_size'], config['action_size']
        self.masked = bool(config.get('support_rllib_mask', False))
        self.observation_space = (_spaces.Dict({'action_mask': _spaces.Box(0, 1, shape=(self.A,)), 'obs': _spaces.Box(-1e5, 1e5, shape=(256,))})
                                  if self.masked else _spaces.Box(-1e5, 1e5, shape=(256,)))
        self.action_space = _spaces.Discrete(self.A)
        self.t = 0
        self.closed = False

    def _obs(self):
        o = np.full((self.B, 256), float(self.t), dtype=np.float32)
        if not self.masked:
            return o
        m = np.ones((self.B, self.A), dtype=np.int64)
        m[:, :self.t + 1] = 0
        return [{'action_mask': m[i], 'obs': o[i]} for i in range(self.B)]

    def seed(self, sd):
        self.sd = sd

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, action):
        assert np.asarray(action).shape == (self.B,)
        self.last_action = np.asarray(action)
        self.t += 1
        return self._obs(), [float(self.t)] * self.B, [int(self.t >= 3)] * self.B, [{} for _ in range(self.B)]

    def render(self):
        pass

    def close(self):
        self.closed = True


@pytest.mark.parametrize('masked', [False, True])
def test_routes_and_json_shapes(masked):
    from rl4rs_amd.server import create_app
    app = create_app(make_env=lambda env_id, config: StubEnv(config))
    c = app.test_client()
    cfg = {'batch_size': 3, 'action_size': 284, 'support_rllib_mask': masked}
    r = c.post('/v1/envs/', json={'env_id': 'SlateRecEnv-v0', 'config': cfg, 'seed': 5})
    assert r.status_code == 200 and set(r.get_json()) == {'instance_id'}
    iid = r.get_json()['instance_id']
    assert len(iid) == 8 and c.get('/v1/envs/').get_json() == {'all_envs': {iid: 'SlateRecEnv-v0'}}
    info = c.get('/v1/envs/%s/action_space/' % iid).get_json()['info']
    assert info == {'name': 'Discrete', 'n': 284}
    oi = c.get('/v1/envs/%s/observation_space/' % iid).get_json()['info']
    if masked:
        assert oi['name'] == 'Dict' and oi['keys'] == ['action_mask', 'obs']
        assert oi['obs']['shape'] == [256] and len(oi['obs']['low']) == 256 and oi['obs']['low'][0] == -1e5
        assert oi['action_mask']['shape'] == [284] and oi['action_mask']['high'][0] == 1.0
    else:
        assert oi['name'] == 'Box' and oi['shape'] == [256] and len(oi['high']) == 256
    obs = c.post('/v1/envs/%s/reset/' % iid).get_json()['observation']
    if masked:                                             # Dict.to_jsonable: one list per key over the batch
        assert set(obs) == {'action_mask', 'obs'} and np.asarray(obs['obs']).shape == (3, 256) and np.asarray(obs['action_mask']).shape == (3, 284)
    else:
        assert np.asarray(obs).shape == (3, 256)
    r = c.post('/v1/envs/%s/step/' % iid, json={'action': [1, 2, 3], 'render': False}).get_json()
    assert set(r) == {'observation', 'reward', 'done', 'info'}
    assert r['reward'] == [1.0, 1.0, 1.0] and r['done'] == [0, 0, 0] and r['info'] == [{}, {}, {}]
    assert c.get('/v1/envs/%s/action_space/contains/283' % iid).get_json() == {'member': True}
    assert c.get('/v1/envs/%s/action_space/contains/284' % iid).get_json() == {'member': False}
    assert 0 <= c.get('/v1/envs/%s/action_space/sample' % iid).get_json()['action'] < 284
    # errors: unknown instance / missing parameter -> 400 + {"message": ...} (gymHttpServer.py:197-236)
    r = c.post('/v1/envs/deadbeef/reset/')
    assert r.status_code == 400 and 'unknown' in r.get_json()['message']
    r = c.post('/v1/envs/%s/step/' % iid, json={'render': False})
    assert r.status_code == 400 and "'action'" in r.get_json()['message']
    env = app.envs.envs[iid]
    assert c.post('/v1/envs/%s/close/' % iid).status_code == 204 and env.closed and c.get('/v1/envs/').get_json() == {'all_envs': {}}


@pytest.mark.parametrize('masked', [False, True])
def test_http_env_round_trip_over_the_wire_format(masked):
    """HttpEnv (httpEnv.py:9-44) on one side, the server on the other: spaces rebuilt from the info replies, observations
    back as arrays / per-env dicts, actions as lists."""
    from rl4rs_amd.server import create_app, HttpEnv
    app = create_app(make_env=lambda env_id, config: StubEnv(config))
    cfg = {'batch_size': 4, 'action_size': 284, 'support_rllib_mask': masked, 'remote_base': ''}
    env = HttpEnv('SlateRecEnv-v0', cfg, session=app.test_client())
    assert env.action_space.n == 284
    obs = env.reset()
    if masked:
        assert isinstance(obs, list) and len(obs) == 4 and obs[0]['obs'].shape == (256,) and obs[0]['action_mask'].shape == (284,)
        assert obs[0]['action_mask'][0] == 0 and obs[0]['action_mask'][1] == 1
        assert tuple(env.observation_space.spaces['obs'].shape) == (256,)
    else:
        assert obs.shape == (4, 256) and tuple(env.observation_space.shape) == (256,)
    obs, reward, done, info = env.step(np.array([5, 6, 7, 8]))
    assert reward == [1.0] * 4 and done == [0] * 4
    assert app.envs.envs[env.instance_id].last_action.tolist() == [5, 6, 7, 8]
    o = obs[0]['obs'] if masked else obs[0]
    assert float(o[0]) == 1.0
    env.close()
    assert app.envs.envs == {}


def test_action_space_contains_delegates_to_the_space():
    """gymHttpServer.py:107-109: member = env.action_space.contains(int(x)) - a Box (continuous-action env) answers for itself
    (a scalar is never a member of a 32-d Box), a Discrete checks the range."""
    from rl4rs_amd.server import create_app

    class ContiStub(StubEnv):
        def __init__(self, config):
            StubEnv.__init__(self, config)
            self.action_space = _spaces.Box(-1, 1, shape=(32,))

    app = create_app(make_env=lambda env_id, config: ContiStub(config))
    c = app.test_client()
    iid = c.post('/v1/envs/', json={'env_id': 'SlateRecEnv-v0', 'config': {'batch_size': 2, 'action_size': 284}}).get_json()['instance_id']
    assert c.get('/v1/envs/%s/action_space/contains/0' % iid).get_json() == {'member': False}
    assert _spaces.Box(-1, 1, shape=(2,)).contains([0.5, -1.0]) and not _spaces.Box(-1, 1, shape=(2,)).contains([0.5, 1.5])
    assert _spaces.Discrete(4).contains(3) and not _spaces.Discrete(4).contains(4) and not _spaces.Discrete(4).contains(1.0)
