"""Offline (logged-policy) dataset generation on device — SURVEY §8 row f1.

Mirrors ``data_generate_rl4rs_a`` / ``_a_conti`` / ``_b`` of the reference (``script/batchrl_trainer.py:172-
374``): roll the env with ``offline_action`` and emit, per episode, ``max_steps + 1`` transitions of
(observation ``[256 + page_items + 1]`` = obs | previous actions | cur_step, logged action, logged reward,
terminal) — the arrays a d3rlpy ``MDPDataset`` is built from.  Everything stays in HBM; only the final arrays are
handed back (torch tensors, or numpy with ``to_numpy=True``).
"""
import numpy as np
import torch


def generate_offline_dataset(env, epochs, shuffle=True, to_numpy=False):
    """env: ``RecEnvBase`` over ``SlateRecEnv``/``SeqSlateRecEnv`` built with ``support_d3rl_mask=True`` and
    ``return_tensors=True``.  Returns dict(observations [N, D], actions [N, 1] or [N, E], rewards [N], terminals [N])
    with N = epochs * batch_size * (max_steps + 1), laid out exactly like the reference (epoch-major, then env, then
    step; epochs permuted with ``np.random.permutation`` when ``shuffle``)."""
    cfg = env.config
    assert cfg.get('support_d3rl_mask', False) and cfg.get('return_tensors', False), \
        "offline generation needs config['support_d3rl_mask'] and config['return_tensors']"
    B, T = cfg['batch_size'], cfg['max_steps']
    conti = bool(cfg.get('support_conti_env', False))
    S = T + 1
    obs0 = env.reset()
    D = obs0.shape[1]
    dev = obs0.device
    E = cfg['action_emb_size'] if conti else 1
    observations = torch.zeros((epochs, B, S, D), dtype=torch.float32, device=dev)
    actions = torch.zeros((epochs, B, S, E), dtype=torch.float32, device=dev)
    rewards = torch.zeros((epochs, B, S), dtype=torch.float32, device=dev)
    terminals = torch.zeros((epochs, B, S), dtype=torch.float32, device=dev)
    for i in range(epochs):
        obs = obs0 if i == 0 else env.reset()
        observations[i, :, 0] = obs.to(torch.float32)
        action = env.offline_action
        actions[i, :, 0] = action.reshape(B, E).to(torch.float32)
        for j in range(T):
            obs, reward, done, info = env.step(action)
            observations[i, :, j + 1] = obs.to(torch.float32)
            action = env.offline_action                       # past the horizon: 0 / action_emb[0] (slate.py:157-161)
            actions[i, :, j + 1] = action.reshape(B, E).to(torch.float32)
            r = env.offline_reward
            if isinstance(r, torch.Tensor):
                rewards[i, :, j + 1] = r.to(torch.float32)
            terminals[i, :, j + 1] = float(done[0])
    if shuffle:
        p = torch.from_numpy(np.random.permutation(epochs)).to(dev)
        observations, actions, rewards, terminals = observations[p], actions[p], rewards[p], terminals[p]
    out = dict(observations=observations.reshape(epochs * B * S, D),
               actions=actions.reshape(epochs * B * S, E),
               rewards=rewards.reshape(epochs * B * S),
               terminals=terminals.reshape(epochs * B * S))
    if to_numpy:
        out = dict((k, v.cpu().numpy()) for k, v in out.items())
    return out
