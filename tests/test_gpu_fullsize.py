"""BASELINE.json full size (SlateRecEnv-v0, B=4096, 284 items, 9 slots, hash 100000): integer state bit-exact vs the
numpy oracle on the whole batch; observations / rewards vs the fp64 DIEN oracle on a random subset of envs (rows are
independent, so a subset is a valid check of the full-size launch geometry); duplicate-record invariance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_full_size_slate_episode(tmp_path):
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from oracle.state import OracleState
    from oracle.dien import OracleDien
    from oracle.env import reward_from_probs
    B, T = 4096, 9
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=1234)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(B - 64, seed=1000, illegal_frac=0.05, special_ids=synth.special_ids_from_text(text))
    recs = recs + recs[:64]                         # 64 duplicated records at the end of the batch
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 100000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'),
           "iteminfo_file": os.path.join(d, 'c.csv'), "is_eval": True, "cache_size": B, "model_seed": 7,
           "return_tensors": True, "support_rllib_mask": True}
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    obs = env.reset(reset_file=True)
    st = OracleState(cfg, recs)
    w = init_dien_weights(cfg, seed=7)
    orc = OracleDien(w, cfg, np.float64)
    pick = np.sort(np.random.RandomState(0).choice(B, 24, replace=False))

    def check_obs(o):
        seq, dense, cat = st.features()
        ref = orc.obs(seq[pick], dense[pick], cat[pick])
        got = o['obs'][torch.from_numpy(pick).cuda()].cpu().numpy()
        assert np.abs(got - ref).max() < 5e-5
        assert np.array_equal(o['action_mask'].cpu().numpy(), st.obs_action_mask())
        assert torch.equal(o['obs'][:64], o['obs'][B - 64:])          # duplicated records -> identical rows

    check_obs(obs)
    for t in range(T):
        a = env.offline_action
        assert np.array_equal(a.cpu().numpy(), np.asarray(st.offline_action))
        obs, reward, done, info = env.step(a)
        st.act(a.cpu().numpy())
        check_obs(obs)
    assert np.array_equal(env.samples.prev_actions, st.prev_actions)
    assert np.array_equal(env.samples.get_violation(), st.get_violation())
    assert np.array_equal(np.asarray(env.offline_reward.cpu()), np.asarray(st.offline_reward))
    # rewards of the picked envs against the fp64 scorer; structural properties on the whole batch
    cs, cd, cc = st.complete_features()
    rows = (pick[:, None] * 9 + np.arange(9)[None, :]).reshape(-1)
    probs = np.zeros((B, 9), dtype=np.float32)
    probs[pick] = orc.prob(cs[rows], cd[rows], cc[rows]).reshape(-1, 9)
    ref = np.asarray(reward_from_probs(st, probs))
    r = reward.cpu().numpy()
    assert np.allclose(r[pick], ref[pick], rtol=1e-5, atol=1e-5)
    viol = st.get_violation()
    assert (r[viol == 0] == 0).all() and (r[viol == 1] > 0).all()
    assert (r <= st.get_price(st.prev_actions).sum(1) + 1e-9).all()
    assert np.array_equal(r[:64], r[B - 64:])
    assert 0.03 < (viol == 0).mean() < 0.12                            # ~5 % injected illegal records


def _full_cfg(d, B, T, recs, **extra):
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 100000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'),
           "iteminfo_file": os.path.join(d, 'c.csv'), "is_eval": True, "cache_size": B, "model_seed": 7,
           "return_tensors": True}
    cfg.update(extra)
    return cfg


def test_full_size_seqslate_t32_episode_then_ppo_iteration(tmp_path):
    """BASELINE configs[2]: SeqSlateRecEnv-v0, B=4096, 32-step horizon (pages of 9: rewards after steps 9, 18, 27; the last 5
    steps never pay, seqslate.py:138).  Integer state of the WHOLE batch bit-exact vs the oracle at every step, observations
    and page rewards of a 24-env subset vs the fp64 DIEN oracle, then one PPO train call of the on-device loop on the same env
    (script/modelfree_train.py:42-44,179-247)."""
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.env.seqslate import SeqSlateRecEnv, SeqSlateState
    from rl4rs_amd.train import Trainer
    from oracle.state import OracleState
    from oracle.dien import OracleDien
    from oracle.env import reward_from_probs, is_reward_step
    B, T = 4096, 32
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=1234)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(B, pages=4, seed=1001, illegal_frac=0.05, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    cfg = _full_cfg(d, B, T, recs, support_rllib_mask=True)
    env = rl4rs_amd.make('SeqSlateRecEnv-v0', recsim=SeqSlateRecEnv(cfg, state_cls=SeqSlateState))
    obs = env.reset(reset_file=True)
    st = OracleState(cfg, recs, seq=True)
    orc = OracleDien(init_dien_weights(cfg, seed=7), cfg, np.float64)
    pick = np.sort(np.random.RandomState(1).choice(B, 24, replace=False))
    pick_t = torch.from_numpy(pick).cuda()

    def check_obs(o, full):
        assert np.array_equal(o['action_mask'].cpu().numpy(), st.obs_action_mask())
        if full:
            seq, dense, cat = st.features()
            ref = orc.obs(seq[pick], dense[pick], cat[pick])
            assert np.abs(o['obs'][pick_t].cpu().numpy() - ref).max() < 5e-5

    check_obs(obs, True)
    n_reward_steps = 0
    for t in range(T):
        a = env.offline_action
        assert np.array_equal(a.cpu().numpy(), np.asarray(st.offline_action))
        obs, reward, done, info = env.step(a)
        st.act(a.cpu().numpy())
        # observations against the fp64 scorer on the first page, at every page boundary and at the end (each costs 24
        # fp64 DIEN rows on the host); masks and integer state at EVERY step
        check_obs(obs, t < 10 or t % 9 in (0, 8) or t == T - 1)
        assert np.array_equal(env.samples.prev_actions, st.prev_actions) if t % 8 == 7 or t == T - 1 else True
        r = reward.cpu().numpy()
        if is_reward_step(st):
            n_reward_steps += 1
            cs, cd, cc = st.complete_features()
            rows = (pick[:, None] * 9 + np.arange(9)[None, :]).reshape(-1)
            probs = np.zeros((B, 9), dtype=np.float32)
            probs[pick] = orc.prob(cs[rows], cd[rows], cc[rows]).reshape(-1, 9)
            ref = np.asarray(reward_from_probs(st, probs))
            assert np.allclose(r[pick], ref[pick], rtol=1e-5, atol=1e-5), t
            viol = st.get_violation()
            assert (r[viol == 0] == 0).all()          # mask mode: violation zeroes the page reward (seqslate.py:154-157)
            assert np.array_equal(np.asarray(env.offline_reward.cpu()), np.asarray(st.offline_reward))
        else:
            assert (r == 0).all()
        assert done == [1 if t == T - 1 else 0] * B
    assert n_reward_steps == 3
    assert np.array_equal(env.samples.get_violation(), st.get_violation())
    # ---- one PPO train call over this env (rollout of 131 072 samples, 512 minibatches of 256 in the persistent pass)
    tr = Trainer(env, algo='PPO', seed=11, init_seed=3)
    p0 = tr.params().clone()
    out = tr.train_iteration()
    assert np.isfinite([v for v in out.values()]).all(), out
    assert not torch.equal(p0, tr.params())
    assert out['kl_coeff'] in (0.1, 0.2, 0.30000000000000004)
    assert out['kl_mean'] >= 0 and out['episode_reward_mean'] > 0
    # the masked policy only plays legal items: page rewards are never zeroed by the violation rule
    assert env.samples.get_violation().all()
    acts = tr.buf['act'].view(T, B).cpu().numpy()
    assert (acts >= 1).all() and (acts < 284).all()
    tr.close()


def test_full_size_continuous_action_episode(tmp_path):
    """BASELINE configs[4], its 1-GPU half: SlateRecEnv-v0 with continuous 32-d actions resolved by the masked float64 K-NN,
    B=4096.  Chosen items / integer state bit-exact vs the oracle on the whole batch (actions = logged items' embeddings
    plus noise, so the K-NN and its masks decide), observations and rewards of a 24-env subset vs the fp64 DIEN oracle."""
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from oracle.state import OracleState
    from oracle.dien import OracleDien
    from oracle.env import reward_from_probs
    B, T = 4096, 9
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=1234)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(B, seed=1002, illegal_frac=0.05, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    cfg = _full_cfg(d, B, T, recs, support_conti_env=True)
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    obs = env.reset(reset_file=True)
    st = OracleState(cfg, recs)
    orc = OracleDien(init_dien_weights(cfg, seed=7), cfg, np.float64)
    pick = np.sort(np.random.RandomState(2).choice(B, 24, replace=False))
    pick_t = torch.from_numpy(pick).cuda()
    rs = np.random.RandomState(5)
    for t in range(T):
        a = env.offline_action                                        # [B, 32] float64 embeddings of the logged items
        assert np.array_equal(a.cpu().numpy(), np.asarray(st.offline_action))
        noisy = a.cpu().numpy() + 0.3 * rs.normal(size=(B, 32))       # off the catalogue points: neighbours compete
        obs, reward, done, info = env.step(torch.from_numpy(noisy).cuda())
        chosen = st.act(noisy)
        assert np.array_equal(env.samples.last_actions.cpu().numpy(), chosen), t
        seq, dense, cat = st.features()
        ref = orc.obs(seq[pick], dense[pick], cat[pick])
        assert np.abs(obs[pick_t].cpu().numpy() - ref).max() < 5e-5
    assert np.array_equal(env.samples.prev_actions, st.prev_actions)
    assert np.array_equal(env.samples.action_mask, st.action_mask)
    assert np.array_equal(env.samples.special_mask, st.special_mask)
    cs, cd, cc = st.complete_features()
    rows = (pick[:, None] * 9 + np.arange(9)[None, :]).reshape(-1)
    probs = np.zeros((B, 9), dtype=np.float32)
    probs[pick] = orc.prob(cs[rows], cd[rows], cc[rows]).reshape(-1, 9)
    ref = np.asarray(reward_from_probs(st, probs))
    r = reward.cpu().numpy()
    assert np.allclose(r[pick], ref[pick], rtol=1e-5, atol=1e-5)
    viol = st.get_violation()
    assert np.array_equal(env.samples.get_violation(), viol)
    assert (r[viol == 0] == 0).all()
    # the masked K-NN can only return legal, unused items: no slate violates the rules
    assert viol.all()
