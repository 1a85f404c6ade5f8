"""Client side of the wire format: ``Client`` (rl4rs/server/gymHttpClient.py) and the gym-style ``HttpEnv``
(rl4rs/server/httpEnv.py:9-44) that RLlib scripts register as ``HttpEnv-v0`` (script/modelfree_train.py:67).
``session`` lets a test drive a Flask test client instead of a socket."""
import json

import numpy as np

from ..env.base import _spaces, _Env


class ServerError(Exception):
    def __init__(self, message, status_code=None):
        Exception.__init__(self, message)
        self.message, self.status_code = message, status_code


class Client(object):
    def __init__(self, remote_base, session=None):
        self.remote_base = remote_base.rstrip('/')
        if session is None:
            import requests
            session = requests.Session()
            session.headers.update({'Content-type': 'application/json'})
        self.session = session

    def _parse(self, resp):
        status = getattr(resp, 'status_code', 200)
        try:
            j = resp.get_json() if hasattr(resp, 'get_json') else resp.json()
        except Exception:
            j = None
        if status not in (200, 204):
            raise ServerError((j or {}).get('message', 'HTTP %d' % status), status)
        return j or {}

    def _post(self, route, data):
        return self._parse(self.session.post(self.remote_base + route, data=json.dumps(data), headers={'Content-type': 'application/json'}))

    def _get(self, route):
        return self._parse(self.session.get(self.remote_base + route))

    def env_create(self, env_id, config={}):
        return self._post('/v1/envs/', {'env_id': env_id, 'config': config})['instance_id']

    def env_list_all(self):
        return self._get('/v1/envs/')['all_envs']

    def env_reset(self, instance_id):
        return self._post('/v1/envs/{}/reset/'.format(instance_id), None)['observation']

    def env_step(self, instance_id, action, render=False):
        r = self._post('/v1/envs/{}/step/'.format(instance_id), {'action': action, 'render': render})
        return [r['observation'], r['reward'], r['done'], r['info']]

    def env_action_space_info(self, instance_id):
        return self._get('/v1/envs/{}/action_space/'.format(instance_id))['info']

    def env_action_space_sample(self, instance_id):
        return self._get('/v1/envs/{}/action_space/sample'.format(instance_id))['action']

    def env_action_space_contains(self, instance_id, x):
        return self._get('/v1/envs/{}/action_space/contains/{}'.format(instance_id, x))['member']

    def env_observation_space_info(self, instance_id):
        return self._get('/v1/envs/{}/observation_space/'.format(instance_id))['info']

    def env_close(self, instance_id):
        self._post('/v1/envs/{}/close/'.format(instance_id), None)


def from_jsonable(space, obs):
    """``space.from_jsonable``: Box -> float32 array over the batch; Dict -> list (one dict per env) of arrays."""
    if space.__class__.__name__.lstrip('_') == 'Dict':
        keys = list(space.spaces.keys())
        cols = dict((k, np.asarray(obs[k], dtype=np.float32)) for k in keys)
        n = len(cols[keys[0]])
        return [dict((k, cols[k][i]) for k in keys) for i in range(n)]
    return np.asarray(obs, dtype=np.float32)


class HttpEnv(_Env):
    metadata = {'render.modes': ['human']}

    def __init__(self, env_id, config={}, session=None):
        self.client = Client(config["remote_base"], session=session)
        self.instance_id = self.client.env_create(env_id, config)
        action_info = self.client.env_action_space_info(self.instance_id)
        obs_info = self.client.env_observation_space_info(self.instance_id)
        if action_info['name'] == 'Box':
            self.action_space = _spaces.Box(np.array(action_info['low']), np.array(action_info['high']), shape=tuple(action_info['shape']))
        else:
            self.action_space = _spaces.Discrete(action_info['n'])
        if obs_info['name'] == 'Box':
            self.observation_space = _spaces.Box(np.array(obs_info['low']), np.array(obs_info['high']), shape=tuple(obs_info['shape']))
        elif obs_info['name'] == 'Dict':
            self.observation_space = _spaces.Dict(dict(
                (k, _spaces.Box(np.array(obs_info[k]['low']).reshape(obs_info[k]['shape']), np.array(obs_info[k]['high']).reshape(obs_info[k]['shape']),
                                shape=tuple(obs_info[k]['shape']))) for k in obs_info['keys']))
        else:
            raise AssertionError("observation space %r is neither Box nor Dict" % (obs_info['name'],))

    def seed(self, sd=0):
        pass

    def step(self, action):
        if isinstance(action, np.ndarray):
            action = action.tolist()
        if isinstance(action, np.integer):
            action = int(action)
        observation, reward, done, info = self.client.env_step(self.instance_id, action, False)
        return from_jsonable(self.observation_space, observation), reward, done, info

    def reset(self):
        return from_jsonable(self.observation_space, self.client.env_reset(self.instance_id))

    def render(self, mode='human', close=False):
        return ''

    def close(self):
        return self.client.env_close(self.instance_id)
