"""Flask app with the reference's routes (rl4rs/server/gymHttpServer.py:239-420, a variant of openai/gym-http-api):

    POST /v1/envs/                                   {env_id, config[, seed]}      -> {instance_id}
    GET  /v1/envs/                                                                  -> {all_envs}
    POST /v1/envs/<id>/reset/                                                       -> {observation}
    POST /v1/envs/<id>/step/                         {action[, render]}            -> {observation, reward, done, info}
    GET  /v1/envs/<id>/action_space/                                                -> {info}
    GET  /v1/envs/<id>/action_space/sample                                          -> {action}
    GET  /v1/envs/<id>/action_space/contains/<x>                                    -> {member}
    GET  /v1/envs/<id>/observation_space/                                           -> {info}
    POST /v1/envs/<id>/observation_space/contains    {name, shape, ...}            -> {member}
    POST /v1/envs/<id>/close/                                                       -> 204

Observations travel as ``observation_space.to_jsonable(obs)`` (Box: nested lists; Dict: one list per key over the batch),
errors as ``{"message": ...}`` with status 400 - what ``HttpEnv`` / ``Client`` on the other side expect.  ``config`` is the
reference's env config dict; the env is built by ``rl4rs_amd.make`` on the server's GPU (numpy-returning mode: the wire is
JSON anyway)."""
import time
import uuid

import numpy as np


class InvalidUsage(Exception):
    status_code = 400

    def __init__(self, message, status_code=None, payload=None):
        Exception.__init__(self)
        self.message = message
        if status_code is not None:
            self.status_code = status_code
        self.payload = payload

    def to_dict(self):
        rv = dict(self.payload or ())
        rv['message'] = self.message
        return rv


def _default_make_env(env_id, config):
    import rl4rs_amd
    config = dict(config)
    config.pop('return_tensors', None)                 # the wire is JSON: numpy / list returns
    if env_id == 'SlateRecEnv-v0':
        from rl4rs_amd.env.slate import SlateRecEnv, SlateState
        return rl4rs_amd.make(env_id, recsim=SlateRecEnv(config, state_cls=SlateState))
    if env_id == 'SeqSlateRecEnv-v0':
        from rl4rs_amd.env.seqslate import SeqSlateRecEnv, SeqSlateState
        return rl4rs_amd.make(env_id, recsim=SeqSlateRecEnv(config, state_cls=SeqSlateState))
    raise InvalidUsage("Attempted to look up malformed environment ID '{}'".format(env_id))


def space_properties(space):
    """gymHttpServer.py:143-171: name + shape/low/high (Box), n (Discrete), keys + per-key boxes (Dict)."""
    info = {'name': space.__class__.__name__.lstrip('_')}

    def box(sp):
        shape = [int(x) for x in sp.shape]
        return {'shape': shape,
                'low': [(float(x) if x != -np.inf else -1e100) for x in np.broadcast_to(np.asarray(sp.low, dtype=np.float64), shape).flatten()],
                'high': [(float(x) if x != +np.inf else +1e100) for x in np.broadcast_to(np.asarray(sp.high, dtype=np.float64), shape).flatten()]}
    if info['name'] == 'Discrete':
        info['n'] = int(space.n)
    elif info['name'] == 'Box':
        info.update(box(space))
    elif info['name'] == 'Dict':
        info['keys'] = [str(k) for k in space.spaces.keys()]
        for k in info['keys']:
            info[k] = box(space.spaces[k])
    return info


def to_jsonable(space, obs):
    """``space.to_jsonable(obs)`` for a BATCH of observations (gym: Box -> nested lists; Dict -> {key: list over the batch})."""
    name = space.__class__.__name__.lstrip('_')
    if name == 'Dict':
        if isinstance(obs, dict):                       # already batched per key
            return dict((k, np.asarray(v).tolist()) for k, v in obs.items())
        return dict((k, [np.asarray(o[k]).tolist() for o in obs]) for k in space.spaces.keys())
    return np.asarray(obs).tolist()


class Envs(object):
    """Container of the env instances of a server (gymHttpServer.py:27-188)."""

    def __init__(self, make_env=None, idle_seconds=300):
        self.envs, self.env_ids, self.env_lasttime = {}, {}, {}
        self.id_len = 8
        self.make_env = make_env or _default_make_env
        self.idle_seconds = idle_seconds

    def _lookup_env(self, instance_id):
        try:
            return self.envs[instance_id]
        except KeyError:
            raise InvalidUsage('Instance_id {} unknown'.format(instance_id))

    def create(self, env_id, config, seed=None):
        for iid in list(self.envs.keys())[:-2]:         # the reference retires instances idle for 5 minutes, keeping the last two
            if abs(time.time() - self.env_lasttime.get(iid, 0)) >= self.idle_seconds:
                self.env_close(iid)
        env = self.make_env(env_id, config)
        if seed:
            env.seed(seed)
        instance_id = str(uuid.uuid4().hex)[:self.id_len]
        self.envs[instance_id], self.env_ids[instance_id] = env, env_id
        self.env_lasttime[instance_id] = time.time()
        return instance_id

    def list_all(self):
        return dict(self.env_ids)

    def reset(self, instance_id):
        env = self._lookup_env(instance_id)
        return to_jsonable(env.observation_space, env.reset())

    def step(self, instance_id, action, render=False):
        self.env_lasttime[instance_id] = time.time()
        env = self._lookup_env(instance_id)
        nice_action = action if isinstance(action, int) else np.array(action)
        if render:
            env.render()
        observation, reward, done, info = env.step(nice_action)
        if isinstance(reward, np.ndarray):
            reward = reward.tolist()
        return [to_jsonable(env.observation_space, observation), reward, done, info]

    def action_space_info(self, instance_id):
        return space_properties(self._lookup_env(instance_id).action_space)

    def observation_space_info(self, instance_id):
        return space_properties(self._lookup_env(instance_id).observation_space)

    def action_space_sample(self, instance_id):
        sp = self._lookup_env(instance_id).action_space
        if sp.__class__.__name__.lstrip('_') == 'Discrete':
            return int(np.random.randint(sp.n))
        return np.random.uniform(np.asarray(sp.low, dtype=np.float64), np.asarray(sp.high, dtype=np.float64), size=tuple(sp.shape)).tolist()

    def action_space_contains(self, instance_id, x):
        # gymHttpServer.py:107-109: env.action_space.contains(int(x)) - the space decides (a Box answers False for a scalar)
        return bool(self._lookup_env(instance_id).action_space.contains(int(x)))

    def observation_space_contains(self, instance_id, j):
        info = self.observation_space_info(instance_id)
        import json
        return all(json.dumps(info.get(k)) == json.dumps(v) for k, v in j.items())

    def env_close(self, instance_id):
        self._lookup_env(instance_id).close()
        del self.envs[instance_id]
        del self.env_ids[instance_id]


def _param(json_body, name, default=None, required=False):
    if json_body is None:
        raise InvalidUsage("Request is not a valid json")
    value = json_body.get(name, None)
    if value is None or value == '' or value == []:
        if required:
            raise InvalidUsage("A required request parameter '{}' was not provided".format(name))
        return default
    return value


def create_app(make_env=None):
    """The Flask application (``app`` of gymHttpServer.py:191-193).  ``make_env(env_id, config)`` overrides how an instance
    is built (tests inject a stub env; the default builds the device env)."""
    from flask import Flask, request, jsonify
    app = Flask('rl4rs_gym_http_server')
    envs = Envs(make_env)
    app.envs = envs

    @app.errorhandler(InvalidUsage)
    def handle_invalid_usage(error):
        response = jsonify(error.to_dict())
        response.status_code = error.status_code
        return response

    @app.route('/v1/envs/', methods=['POST'])
    def env_create():
        j = request.get_json()
        instance_id = envs.create(_param(j, 'env_id', required=True), _param(j, 'config', required=True), _param(j, 'seed'))
        return jsonify(instance_id=instance_id)

    @app.route('/v1/envs/', methods=['GET'])
    def env_list_all():
        return jsonify(all_envs=envs.list_all())

    @app.route('/v1/envs/<instance_id>/reset/', methods=['POST'])
    def env_reset(instance_id):
        return jsonify(observation=envs.reset(instance_id))

    @app.route('/v1/envs/<instance_id>/step/', methods=['POST'])
    def env_step(instance_id):
        j = request.get_json()
        obs, reward, done, info = envs.step(instance_id, _param(j, 'action', required=True), _param(j, 'render', False))
        return jsonify(observation=obs, reward=reward, done=done, info=info)

    @app.route('/v1/envs/<instance_id>/action_space/', methods=['GET'])
    def env_action_space_info(instance_id):
        return jsonify(info=envs.action_space_info(instance_id))

    @app.route('/v1/envs/<instance_id>/action_space/sample', methods=['GET'])
    def env_action_space_sample(instance_id):
        return jsonify(action=envs.action_space_sample(instance_id))

    @app.route('/v1/envs/<instance_id>/action_space/contains/<x>', methods=['GET'])
    def env_action_space_contains(instance_id, x):
        return jsonify(member=envs.action_space_contains(instance_id, x))

    @app.route('/v1/envs/<instance_id>/observation_space/', methods=['GET'])
    def env_observation_space_info(instance_id):
        return jsonify(info=envs.observation_space_info(instance_id))

    @app.route('/v1/envs/<instance_id>/observation_space/contains', methods=['POST'])
    def env_observation_space_contains(instance_id):
        return jsonify(member=envs.observation_space_contains(instance_id, request.get_json()))

    @app.route('/v1/envs/<instance_id>/close/', methods=['POST'])
    def env_close(instance_id):
        envs.env_close(instance_id)
        return ('', 204)

    return app


def main(argv=None):
    """``python -m rl4rs_amd.server.gym_http_server -l 127.0.0.1 -p 5000`` (gymHttpServer.py:455-462)."""
    import argparse
    ap = argparse.ArgumentParser(description='Start a rl4rs HTTP env server over the device env')
    ap.add_argument('-l', '--listen', default='127.0.0.1')
    ap.add_argument('-p', '--port', default=5000, type=int)
    a = ap.parse_args(argv)
    create_app().run(host=a.listen, port=a.port, threaded=False)       # one env handle = one stream = not thread-safe


if __name__ == '__main__':
    main()
