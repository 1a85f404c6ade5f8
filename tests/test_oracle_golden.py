"""The numpy oracle must reproduce the reference state machine bit-for-bit on the golden fixtures
(captured from rl4rs.env.slate.SlateState / seqslate.SeqSlateState by tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle.env import reward_from_probs, is_reward_step
from oracle.state import OracleState, nearest_neighbor
from helpers import SCENARIOS, load_scenario, golden_equal, golden_has


@pytest.mark.parametrize('name', SCENARIOS)
def test_oracle_matches_reference(name):
    m, cfg, records, g = load_scenario(name)
    st = OracleState(cfg, records, seq=m['seq'])
    T = cfg['max_steps']
    seq, dense, cat = st.features()
    assert np.array_equal(seq, g['seq_init'])
    assert golden_equal(g, 'dense_init', dense) and dense.dtype == np.float32
    assert np.array_equal(cat, g['cat_init'])
    assert np.array_equal(st.obs_action_mask(), g['obsmask_init'])
    assert np.array_equal(st.action_emb, g['action_emb'])
    assert list(st.user) == list(g['user'])
    for t in range(T):
        off = np.asarray(st.offline_action)
        assert np.array_equal(off, g['offline_action_%d' % t]), t
        st.act(g['action_in_%d' % t])
        assert np.array_equal(st.prev_actions, g['prev_actions_%d' % t]), t
        assert np.array_equal(st.action_mask, g['action_mask_%d' % t]), t
        assert np.array_equal(st.special_mask, g['special_mask_%d' % t]), t
        seq, dense, cat = st.features()
        assert np.array_equal(seq, g['seq_%d' % t]), t
        assert golden_equal(g, 'dense_%d' % t, dense), t
        assert np.array_equal(cat, g['cat_%d' % t]), t
        assert np.array_equal(st.obs_action_mask(), g['obsmask_%d' % t]), t
        pa, cur = st.masked_actions()
        assert np.array_equal(pa, g['d3rl_prev_%d' % t]), t
        assert np.array_equal(cur, g['d3rl_cur_%d' % t]), t
        reward = np.asarray(reward_from_probs(st, g['probs_%d' % t]), dtype=np.float64)
        assert np.array_equal(reward, g['reward_%d' % t]), t     # bit-exact f64
        if is_reward_step(st):
            cs, cd, cc = st.complete_features()
            assert np.array_equal(cs, g['c_seq_%d' % t]), t
            assert golden_equal(g, 'c_dense_%d' % t, cd), t
            assert np.array_equal(cc, g['c_cat_%d' % t]), t
            assert np.array_equal(st.get_violation(), g['violation_%d' % t]), t
        else:
            assert not golden_has(g, 'c_dense_%d' % t)
        assert np.array_equal(np.asarray(st.offline_reward, dtype=np.float64), g['offline_reward_%d' % t]), t
    assert np.array_equal(np.asarray(st.offline_action), g['offline_action_end'])
    assert np.array_equal(st.get_violation(), g['violation_end'])


def test_tutorial_known_answers():
    """tutorial.ipynb cells 4/10/12 known answers that need no checkpoint."""
    m, cfg, records, g = load_scenario('real_discrete')
    assert np.array_equal(g['knn_ones'], [53, 53, 53, 53])
    st = OracleState(cfg, records)
    assert np.array_equal(nearest_neighbor(np.full((2, 32), 1), st.action_emb), [53, 53])
    for t in range(9):
        st.act(st.offline_action)
    assert st.offline_reward[0] == 162.99999999999997
    assert st.get_price(st.prev_actions)[0].tolist() == [14.6, 12.8, 11.2, 13.5, 19.1, 19.3, 17.3, 26.1, 29.1]
    assert st.get_violation()[0] == 1
    assert st.user == ['1']
    seq, dense, cat = st.features()
    assert cat[0].tolist() == [64054, 50887, 66367, 44932, 59460, 20543, 83978, 50138, 74820, 58670,
                               1, 3, 5, 29, 72, 53, 52, 164, 211, 172, 172]
    assert seq[0, 0, :8].tolist() == [14, 139, 83, 83, 125, 184, 240, 160]
    # simulator_env_test.py:61-63 invariant: env-built dense[32:392] == the record's own item_feature
    item_feature = np.array(list(map(float, records[0].split('@')[7].replace(';', ',').split(','))),
                            dtype=np.float32)
    assert np.array_equal(dense[0, 32:392], item_feature[:360])
    # catalogue facts
    c = st.cat
    assert c.location_mask.sum(1).tolist() == [39, 108, 136, 1]
    assert len(c.special_items) == 113 and min(c.special_items) >= 62
    assert abs(np.linalg.norm(c.action_emb[1]) - 1) < 1e-12
    assert c.price[1:].min() == 7.0 and c.price.max() == 1478.1


@pytest.mark.parametrize('name', ['slate_discrete', 'real_discrete'])
def test_faithful_per_sample_loop_matches_reference(name):
    """oracle/faithful.py (the reference's own control flow: one python iteration per sample, nested-list state; the
    'faithful' cpu_baseline leg of bench.py) against the golden vectors captured from the reference, bit for bit."""
    from oracle.faithful import FaithfulSlateEnv
    m, cfg, records, g = load_scenario(name)
    T = cfg['max_steps']

    class Stub(object):
        def __init__(self):
            self.probs = None

        def obs(self, seq, dense, cat):
            return np.zeros((len(cat), 256), dtype=np.float32)

        def prob(self, seq, dense, cat):
            return np.asarray(self.probs, dtype=np.float32).reshape(-1)

    stub = Stub()
    env = FaithfulSlateEnv(cfg, records, stub)
    seq, dense, cat = env._features(env.state)
    assert np.array_equal(seq, g['seq_init']) and golden_equal(g, 'dense_init', dense) and np.array_equal(cat, g['cat_init'])
    for t in range(T):
        assert np.array_equal(np.asarray(env.offline_action), g['offline_action_%d' % t])
        stub.probs = g['probs_%d' % t]
        obs, reward, done, _ = env.step(g['action_in_%d' % t])
        assert np.array_equal(env.prev_actions, g['prev_actions_%d' % t]), t
        assert np.array_equal(env.action_mask, g['action_mask_%d' % t]), t
        assert np.array_equal(env.special_mask, g['special_mask_%d' % t]), t
        seq, dense, cat = env._features(env.state)
        assert np.array_equal(seq, g['seq_%d' % t]) and golden_equal(g, 'dense_%d' % t, dense) and np.array_equal(cat, g['cat_%d' % t]), t
        assert np.array_equal(np.asarray(reward, dtype=np.float64), g['reward_%d' % t]), t
        assert done == [1 if t == T - 1 else 0] * cfg['batch_size']
    cs, cd, cc = env._features(env.complete_states())
    assert np.array_equal(cs, g['c_seq_%d' % (T - 1)]) and golden_equal(g, 'c_dense_%d' % (T - 1), cd) and np.array_equal(cc, g['c_cat_%d' % (T - 1)])
    assert np.array_equal(env.violation(), g['violation_end'])
