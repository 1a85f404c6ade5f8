"""Ray-free duck type of ``MyVectorEnvWrapper`` (``rl4rs/utils/rllib_vector_env.py:9-78``): wraps ONE batched
env as ``batch_size`` sub-envs with RLlib's ``VectorEnv`` method names."""
import numpy as np


class MyVectorEnvWrapper(object):
    def __init__(self, env, batch_size):
        self.env = env
        self.reset_cache = []
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.num_envs = batch_size

    def vector_reset(self):
        return self.env.reset()

    def reset_at(self, index=None):
        if index == 0:                               # the whole batch resets with sub-env 0
            self.reset_cache = self.env.reset()
        return self.reset_cache[index]

    def vector_step(self, actions):
        return self.env.step(np.array(actions))

    def get_unwrapped(self):
        return [self.env, ] * self.num_envs

    def try_render_at(self, index=None):
        return self.env.render()
