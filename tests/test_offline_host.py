"""Host-side pieces of the offline-RL path that need no GPU: d3rlpy 0.91's episode -> transition rule
(script/batchrl_trainer.py:186-197 stores the reward of an action with the NEXT observation) and the parameter layout of the
learner networks (rl4rs/nets/cql/encoder.py:24-37)."""
import numpy as np
import torch

from rl4rs_amd.offline_rl import init_qnet_params, transitions_from_mdp


def test_transitions_follow_the_d3rlpy_episode_rule():
    obs = np.arange(7 * 3, dtype=np.float32).reshape(7, 3) + 1
    act = np.array([[5], [6], [7], [8], [9], [10], [11]], np.float32)
    rew = np.array([0, 1, 2, 0, 3, 4, 5], np.float32)
    ter = np.array([0, 0, 1, 0, 0, 1, 0], np.float32)            # two complete episodes + a dangling, non-terminal row
    o, a, r, n, t = transitions_from_mdp(obs, act, rew, ter)
    assert o.shape == (6, 3) and a.dtype == torch.int32 and a.tolist() == [5, 6, 7, 8, 9, 10]
    assert r.tolist() == [1, 2, 0, 3, 4, 0]                        # reward of a_t is rewards[t + 1]; terminal rows get 0
    assert t.tolist() == [0, 0, 1, 0, 0, 1]
    assert torch.equal(n[0], torch.from_numpy(obs[1])) and torch.equal(n[3], torch.from_numpy(obs[4]))
    assert float(n[2].abs().sum()) == 0 and float(n[5].abs().sum()) == 0      # successor of a terminal row: zeros
    # a dataset that ends on a terminal row keeps every row
    o2 = transitions_from_mdp(obs[:6], act[:6], rew[:6], ter[:6])[0]
    assert o2.shape[0] == 6
    # the generated datasets: episodes of max_steps + 1 rows, terminal on the last one
    B, S = 4, 10
    obs = np.random.RandomState(0).randn(B * S, 5).astype(np.float32)
    ter = np.tile(np.r_[np.zeros(S - 1), 1.0], B).astype(np.float32)
    rew = np.tile(np.r_[np.zeros(S - 1), 7.0], B).astype(np.float32)
    o, a, r, n, t = transitions_from_mdp(obs, np.zeros((B * S, 1), np.float32), rew, ter)
    assert o.shape[0] == B * S and int(t.sum()) == B
    assert r.reshape(B, S)[:, S - 2].tolist() == [7.0] * B and float(r.reshape(B, S)[:, S - 1].abs().sum()) == 0


def test_qnet_parameter_layout():
    p = init_qnet_params(266, 284, mask_size=10, seed=1)           # CustomVectorEncoder(with_q=True, hidden_units=[256])
    assert p['fc1_w'].shape == (266, 256) and p['emb'].shape == (284, 32)
    assert p['fc2_w'].shape == (256 + 10 * 32, 284) and p['head_w'].shape == (284, 284)
    assert abs(p['fc1_w']).max() <= 1 / np.sqrt(266) and abs(p['head_b']).max() <= 1 / np.sqrt(284)
    q = init_qnet_params(266, 284, mask_size=0, seed=1)            # d3rlpy VectorEncoder [256, 256]
    assert 'emb' not in q and q['fc2_w'].shape == (256, 256) and q['head_w'].shape == (256, 284)
    assert all(v.dtype == np.float32 for v in list(p.values()) + list(q.values()))
