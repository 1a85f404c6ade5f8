"""End-to-end parity through the reference-shaped API: ``SlateRecEnv``/``SeqSlateRecEnv`` + ``RecEnvBase``
on the GPU against the oracle env (numpy state machine + fp64 DIEN) on the same files and weights.

Integer quantities (chosen items, masks, done) must be identical; observations within 5e-5 abs;
rewards within rtol 1e-5 + atol 1e-5 (north star: fp32 rewards within 1e-5)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(tmp_path, seq, B, T, **flags):
    from rl4rs_amd import synth
    from rl4rs_amd.nets.dien import init_dien_weights, save_weights
    d = str(tmp_path)
    cat_path = os.path.join(d, 'item_info.csv')
    cat_text = synth.make_catalog_text(seed=21)
    synth.write_text(cat_path, cat_text)
    records = synth.make_records(B + 5, pages=4 if seq else 1, seed=8, illegal_frac=0.3, hash_size=5000,
                                 special_ids=synth.special_ids_from_text(cat_text))
    log_path = os.path.join(d, 'log.csv')
    synth.write_records(log_path, records)
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 5000, "seq_num": 2, "emb_size": 128,
           "page_items": 9, "hidden_units": 128, "max_steps": T, "action_emb_size": 32,
           "sample_file": log_path, "iteminfo_file": cat_path, "is_eval": True, "cache_size": B}
    cfg.update(flags)
    w = init_dien_weights(cfg, seed=5, emb_scale=0.5, bias_noise=0.2)
    wpath = os.path.join(d, 'dien.npz')
    save_weights(wpath, w)
    cfg['model_file'] = wpath
    return cfg, records, w


def _make(cfg, seq):
    import rl4rs_amd
    if seq:
        from rl4rs.env.seqslate import SeqSlateRecEnv, SeqSlateState
        sim = SeqSlateRecEnv(cfg, state_cls=SeqSlateState)
        return rl4rs_amd.make('SeqSlateRecEnv-v0', recsim=sim)
    from rl4rs.env.slate import SlateRecEnv, SlateState
    sim = SlateRecEnv(cfg, state_cls=SlateState)
    return rl4rs_amd.make('SlateRecEnv-v0', recsim=sim)


def _oracle(cfg, records, w, seq):
    from oracle.dien import OracleDien
    from oracle.env import OracleEnv
    return OracleEnv(cfg, records[:cfg['batch_size']], OracleDien(w, cfg, np.float64), seq=seq)


@pytest.mark.parametrize('seq,T', [(False, 9), (True, 36), (True, 32)])
@pytest.mark.parametrize('mode', ['plain', 'rllib', 'd3rl', 'conti', 'onehot'])
def test_episode_matches_oracle(tmp_path, seq, T, mode):
    # onehot: support_onehot_action (slate.py:22-25; the continuous dataset of script/batchrl_trainer.py:224-225): the action
    # embedding table is eye(284), a continuous action is a 284-d vector the masked K-NN resolves
    flags = {'plain': {}, 'rllib': {'support_rllib_mask': True}, 'd3rl': {'support_d3rl_mask': True},
             'conti': {'support_conti_env': True}, 'onehot': {'support_conti_env': True, 'support_onehot_action': True}}[mode]
    B = 12
    cfg, records, w = _setup(tmp_path, seq, B, T, **flags)
    env = _make(cfg, seq)
    orc = _oracle(cfg, records, w, seq)
    obs = env.reset(reset_file=True)
    o_obs = orc.reset()
    assert list(env.user_id) == list(orc.samples.user)

    def check_obs(obs, o_obs):
        if mode == 'rllib':
            assert isinstance(obs, list) and len(obs) == B and set(obs[0]) == {'action_mask', 'obs'}
            got = np.stack([x['obs'] for x in obs])
            assert np.array_equal(np.stack([x['action_mask'] for x in obs]), o_obs['action_mask'])
            assert obs[0]['action_mask'].dtype == np.int64
        elif mode == 'd3rl':
            assert obs.shape == (B, 256 + o_obs['masked_actions'].shape[1] + 1)
            got = obs[:, :256]
            assert np.array_equal(obs[:, 256:-1], o_obs['masked_actions'])
            assert np.array_equal(obs[:, -1:], o_obs['cur_steps'])
        else:
            got = obs
        assert got.shape == (B, 256)
        assert np.abs(got - o_obs['obs']).max() < 5e-5

    check_obs(obs, o_obs)
    rs = np.random.RandomState(4)
    for t in range(T):
        if mode == 'conti':
            a = rs.randn(B, 32).astype(np.float32)
        elif mode == 'onehot':
            assert env.config['action_emb_size'] == 284 and tuple(env.action_space.shape) == (284,)       # slate.py:23: the state rewrites the config
            off = np.asarray(env.offline_action, dtype=np.float64)
            assert off.shape == (B, 284) and np.array_equal(off, np.asarray(orc.samples.offline_action, dtype=np.float64))
            assert ((off == 0) | (off == 1)).all() and (off.sum(axis=1) == 1).all()         # rows of eye(284)
            a = rs.randn(B, 284).astype(np.float32)
            a[:B // 3] = off[:B // 3]                        # a third of the batch replays the logged item's one-hot row
        else:
            a = env.offline_action
            assert a == orc.samples.offline_action
            if t % 4 == 3:
                # off-policy ids (duplicates / wrong layers -> violations) for the second half of the batch
                a = list(a[:B // 2]) + [int(x) for x in rs.randint(0, 284, size=B - B // 2)]
        obs, reward, done, info = env.step(a)
        o_obs, o_reward, o_done, chosen = orc.step(a)
        check_obs(obs, o_obs)
        assert isinstance(reward, list) and len(reward) == B
        assert np.allclose(reward, o_reward, rtol=1e-5, atol=1e-5), (t, reward, o_reward)
        assert done == o_done and isinstance(done, list)
        assert isinstance(info, list) and len(info) == B
        assert np.array_equal(env.samples.prev_actions, orc.samples.prev_actions)
        assert np.allclose(np.asarray(env.offline_reward, dtype=np.float64),
                           np.asarray(orc.samples.offline_reward, dtype=np.float64), rtol=0, atol=0)
    assert seq or mode in ('conti', 'onehot') or any(r != 0 for r in reward)
    assert np.array_equal(env.samples.get_violation(), orc.samples.get_violation())
    assert np.array_equal(env.samples.action_mask, orc.samples.action_mask)
    assert np.array_equal(env.samples.special_mask, orc.samples.special_mask)


def test_rawstate_and_tensor_modes(tmp_path):
    import torch
    B, T = 6, 9
    cfg, records, w = _setup(tmp_path, False, B, T, support_rllib_mask=True, rawstate_as_obs=True)
    env = _make(cfg, False)
    orc = _oracle(dict(cfg, rawstate_as_obs=True), records, w, False)
    obs = env.reset(reset_file=True)
    o = orc.reset()
    assert set(obs[0]) == {'action_mask', 'category_feature', 'dense_feature', 'sequence_feature'}
    assert obs[0]['sequence_feature'].shape == (2, 64) and len(obs[0]['dense_feature']) == 432
    for t in range(T):
        a = env.offline_action
        obs, reward, done, info = env.step(a)
        o, o_reward, _, _ = orc.step(a)
        assert np.array_equal(np.stack([x['dense_feature'] for x in obs]), o['dense_feature'])
        assert np.array_equal(np.stack([x['category_feature'] for x in obs]), o['category_feature'])
        assert np.array_equal(np.stack([x['sequence_feature'] for x in obs]), o['sequence_feature'])
        assert np.array_equal(np.stack([x['action_mask'] for x in obs]), o['action_mask'])
        assert np.allclose(reward, o_reward, rtol=1e-5, atol=1e-5)
    # zero-copy mode: torch tensors, no host round trip
    cfg2, records, w = _setup(tmp_path, False, B, T, return_tensors=True, simulator_info_fetch=True)
    env2 = _make(cfg2, False)
    orc2 = _oracle(cfg2, records, w, False)
    obs = env2.reset(reset_file=True)
    o = orc2.reset()
    assert isinstance(obs, torch.Tensor) and obs.is_cuda and obs.shape == (B, 256)
    for t in range(T):
        a = env2.offline_action
        assert isinstance(a, torch.Tensor)
        obs, reward, done, info = env2.step(a)
        o, o_reward, _, _ = orc2.step(a.cpu().numpy())
        assert np.abs(obs.cpu().numpy() - o['obs']).max() < 5e-5
        assert isinstance(reward, torch.Tensor) and reward.dtype == torch.float64
        assert np.allclose(reward.cpu().numpy(), o_reward, rtol=1e-5, atol=1e-5)
    assert 'click_p' in info[0] and info[0]['click_p'].shape == (9,)


def test_batch_of_one_and_sampling_semantics(tmp_path):
    """single_elem_support unwrapping (base.py:9-23) and RecDataBase cache/wrap/sampling (base.py:82-108)."""
    cfg, records, w = _setup(tmp_path, False, 1, 9)
    env = _make(cfg, False)
    obs = env.reset(reset_file=True)
    assert obs.shape == (256,)
    a = env.offline_action
    assert isinstance(a, int)
    obs, reward, done, info = env.step(a)
    assert obs.shape == (256,) and reward == 0 and done == 0 and info == {}
    # train-mode sampling draws np.random.choice from the global RNG exactly like the reference
    cfg, records, w = _setup(tmp_path, False, 4, 9, is_eval=False, cache_size=6)
    env = _make(cfg, False)
    env.seed(123)
    env.reset(reset_file=True)
    np.random.seed(123)
    cache = records[:6]
    expect = np.random.choice(cache, 4)
    assert list(env.samples.records) == list(expect)
    # the file has 9 lines + trailing newline: the next reset wraps (base.py:84-88: skip line 0, take line 1)
    env.reset()
    assert env.sim._recData.sample_list == records[6:9] + [records[1]] + records[2:4]


def test_vector_env_wrapper(tmp_path):
    from rl4rs.utils.rllib_vector_env import MyVectorEnvWrapper
    cfg, records, w = _setup(tmp_path, False, 5, 9, support_rllib_mask=True)
    env = _make(cfg, False)
    venv = MyVectorEnvWrapper(env, 5)
    assert venv.num_envs == 5 and len(venv.get_unwrapped()) == 5
    first = venv.reset_at(0)
    assert set(first) == {'action_mask', 'obs'}
    assert venv.reset_at(3)['obs'].shape == (256,)
    obs, rew, done, info = venv.vector_step(env.offline_action)
    assert len(obs) == 5 and len(rew) == 5


def test_custom_state_cls_is_rejected_loudly(tmp_path):
    from rl4rs.env.slate import SlateRecEnv
    from rl4rs.env import RecState
    cfg, records, w = _setup(tmp_path, False, 2, 9)

    class Mine(RecState):
        pass
    with pytest.raises(NotImplementedError):
        SlateRecEnv(cfg, state_cls=Mine)


@pytest.mark.parametrize('seq,T', [(False, 9), (True, 36)])
def test_state_row_reuse_is_bit_identical(tmp_path, seq, T):
    """The reward forward skips the last complete-state row of each env and takes its click probability from the
    state row scored one call earlier: results must be bit-identical to scoring all 9 rows."""
    import torch
    out = []
    for reuse_off in (False, True):
        cfg, records, w = _setup(tmp_path, seq, 10, T, return_tensors=True, simulator_info_fetch=True,
                                 no_state_row_reuse=reuse_off)
        env = _make(cfg, seq)
        env.reset(reset_file=True)
        rewards = []
        for t in range(T):
            obs, reward, done, info = env.step(env.offline_action)
            rewards.append(reward.clone())
        out.append((torch.stack(rewards), np.stack([i['click_p'] for i in info])))
    assert torch.equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert out[0][0].abs().sum() > 0


@pytest.mark.parametrize('seq,T', [(False, 9), (True, 18)])
def test_history_dedup_is_bit_identical(tmp_path, seq, T):
    """Training-mode sampling draws the batch with replacement from a smaller cache window (base.py:92-100), so envs
    share user histories; the scorer encodes each distinct history once.  Must equal one-slot-per-env bit for bit."""
    import torch
    out = []
    for off, no_order in ((False, False), (True, False), (False, True)):
        # third leg: dedup on, but without the slot-sorted processing order (rl4rs_dien_set_row_order): a locality hint only
        cfg, records, w = _setup(tmp_path, seq, 24, T, return_tensors=True, no_history_dedup=off, no_row_order=no_order)
        cfg.update(is_eval=False, cache_size=7)          # 24 envs drawn from a 7-line window
        env = _make(cfg, seq)
        env.seed(123)
        obs = env.reset(reset_file=True)
        rows = list(env.samples.records.rows)
        assert len(set(rows)) < len(rows)
        assert (getattr(env.samples, '_hist_unique', None) is not None)
        trace = [obs.clone()]
        for t in range(T):
            obs, reward, done, info = env.step(env.offline_action)
            trace += [obs.clone(), reward.clone()]
        out.append((rows, trace))
    assert out[0][0] == out[1][0] == out[2][0]
    for a, b, c in zip(out[0][1], out[1][1], out[2][1]):
        assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize('seq,T', [(False, 9), (True, 36)])
def test_offline_dataset_generation_matches_oracle_replay(tmp_path, seq, T):
    """f1: device-side data_generate_rl4rs_* (script/batchrl_trainer.py:172-217) vs the same loop over the oracle env."""
    from rl4rs_amd.offline import generate_offline_dataset
    B = 6
    cfg, records, w = _setup(tmp_path, seq, B, T, support_d3rl_mask=True, return_tensors=True)
    env = _make(cfg, seq)
    env.reset(reset_file=True)          # so that the generator's first reset() reads the SECOND cache window
    ds = generate_offline_dataset(env, epochs=1, shuffle=False, to_numpy=True)
    cfg_o = dict(cfg, return_tensors=False)
    orc = _oracle(cfg_o, records, w, seq)
    # eval mode + cache_size == B: the generator's reset() took lines [B, 2B) -> wraps like base.py:84-88
    lines = records + ['']
    cur, window = B, []
    for _ in range(B):
        t = lines[cur] if cur < len(lines) else ''
        cur += 1
        if not t:
            cur = 2
            t = lines[1]
        window.append(t)
    o = orc.reset(records=window)
    S, P = T + 1, 9
    obs = ds['observations'].reshape(B, S, -1)
    act = ds['actions'].reshape(B, S)
    rew = ds['rewards'].reshape(B, S)
    term = ds['terminals'].reshape(B, S)
    assert obs.shape[2] == 256 + P + 1

    def check(k, o):
        assert np.abs(obs[:, k, :256] - o['obs']).max() < 5e-5
        assert np.array_equal(obs[:, k, 256:-1], o['masked_actions'])
        assert np.array_equal(obs[:, k, -1:], o['cur_steps'])

    check(0, o)
    assert rew[:, 0].tolist() == [0] * B and term[:, 0].tolist() == [0] * B
    for j in range(T):
        a = np.asarray(orc.samples.offline_action)
        assert np.array_equal(act[:, j], a)
        o, _, done, _ = orc.step(a)
        check(j + 1, o)
        assert np.allclose(rew[:, j + 1], np.asarray(orc.samples.offline_reward, dtype=np.float64), rtol=1e-6)
        assert term[:, j + 1].tolist() == [float(x) for x in done]
    assert act[:, T].tolist() == [0] * B
    assert rew[:, T].sum() > 0


@pytest.mark.parametrize('algo', ['dien', 'lstm'])
def test_model_file_may_be_a_tf_checkpoint_prefix(tmp_path, algo):
    """base.py:148-151: ``model_file`` is a ``tf.train.Saver`` prefix in the reference.  The same weights handed over as a
    checkpoint (read without TensorFlow, ``utils.tfckpt``) or as an .npz must give bit-identical episodes; a trainer's
    ``save(prefix)`` is what ``reload_model(prefix)`` takes (supervised_train.py:44-46)."""
    from rl4rs_amd.utils import tfckpt
    from rl4rs_amd.nets import simnets
    B = 64
    cfg, records, w = _setup(tmp_path, False, B, 9, algo=algo)
    if algo != 'dien':
        w = simnets.init_simnet_weights(cfg, algo, seed=5, emb_scale=0.5, bias_noise=0.2)
        np.savez(os.path.join(str(tmp_path), 'sim.npz'), **w)
        cfg['model_file'] = os.path.join(str(tmp_path), 'sim.npz')
    prefix = os.path.join(str(tmp_path), 'simulator_a_' + algo)
    tfckpt.save_simulator_weights(prefix, w, cfg, algo)
    runs = []
    for model_file in (cfg['model_file'], prefix):
        env = _make(dict(cfg, model_file=model_file), False)
        obs = [np.asarray(env.reset(reset_file=True))]
        rewards = []
        for t in range(9):
            o, r, done, _ = env.step(env.offline_action)
            obs.append(np.asarray(o))
            rewards.append(np.asarray(r))
        runs.append((np.stack(obs), np.stack(rewards)))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert np.abs(runs[0][1]).sum() > 0
    with pytest.raises(FileNotFoundError):
        _make(dict(cfg, model_file=os.path.join(str(tmp_path), 'no_such_checkpoint')), False)


@pytest.mark.parametrize('kind', ['slate', 'slate_mask', 'seq', 'conti', 'slate_lstm', 'slate_widedeep'])
def test_fused_step_is_bit_identical(tmp_path, kind):
    """rl4rs_env_step_discrete / rl4rs_env_step_conti (one library call per transition, zero-copy mode) against the composed
    path (act, obs forward, complete rows, reward forward, reward: one call each; config['no_fused_step']): observations,
    rewards, masks and integer state bit for bit over whole episodes, twice (second episode = new batch on the same handles)."""
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from rl4rs_amd.env.seqslate import SeqSlateRecEnv, SeqSlateState
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=4)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    seq = kind == 'seq'
    T = 18 if seq else 9
    recs = synth.make_records(200, pages=2 if seq else 1, seed=3, hash_size=2000, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    B = 48
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'),
           "iteminfo_file": os.path.join(d, 'c.csv'), "cache_size": 128, "model_seed": 3, "return_tensors": True}
    if kind == 'slate_mask' or seq:
        cfg['support_rllib_mask'] = True
    if kind == 'conti':
        cfg['support_conti_env'] = True
    if kind == 'slate_lstm':
        cfg['algo'] = 'lstm'
    if kind == 'slate_widedeep':                  # the family whose observation is wider than 256 (256 + U + Cn * E)
        cfg['algo'] = 'widedeep'

    def run(fused, copies=False):
        c = dict(cfg, no_fused_step=not fused, copy_outputs=copies)
        if seq:
            env = rl4rs_amd.make('SeqSlateRecEnv-v0', recsim=SeqSlateRecEnv(c, state_cls=SeqSlateState))
        else:
            env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(c, state_cls=SlateState))
        env.seed(11)
        out = []
        for ep in range(2):
            env.reset()
            for t in range(T):
                a = env.offline_action
                obs, reward, done, info = env.step(a)
                o = obs['obs'] if isinstance(obs, dict) else obs
                m = obs['action_mask'].clone() if isinstance(obs, dict) else torch.zeros(1)
                out.append((o.clone(), reward.clone(), m, list(done), env.samples.last_actions.clone()))
            out.append((torch.from_numpy(env.samples.prev_actions), torch.from_numpy(env.samples.get_violation())))
        assert (env.sim._step.__func__ is SlateRecEnv._step) and (getattr(env.sim, '_stepper', None) is not None) == fused
        return out

    a, b = run(True), run(False)
    assert len(a) == len(b)
    n_reward = 0
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            if torch.is_tensor(u):
                assert torch.equal(u.cpu(), v.cpu())
            else:
                assert u == v
        if len(x) == 5 and float(x[1].abs().sum()) > 0:
            n_reward += 1
    assert n_reward == (4 if seq else 2)


def test_http_env_round_trip(tmp_path):
    """the device env behind the reference's HTTP wire format (rl4rs_amd.server: routes of
    gymHttpServer.py:239-420, HttpEnv of httpEnv.py:9-44), driven through a Flask test client: same observations, rewards
    and done flags as the in-process env (JSON carries float32 exactly as decimal text -> compare at float32 resolution)."""
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from rl4rs.server.gymHttpServer import create_app
    from rl4rs.server.httpEnv import HttpEnv
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=4)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(40, seed=3, hash_size=2000, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    B, T = 8, 9
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'),
           "iteminfo_file": os.path.join(d, 'c.csv'), "cache_size": B, "is_eval": True, "model_seed": 3,
           "support_rllib_mask": True, "remote_base": ""}
    app = create_app()
    henv = HttpEnv('SlateRecEnv-v0', cfg, session=app.test_client())
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(dict(cfg), state_cls=SlateState))
    assert henv.action_space.n == 284 and sorted(henv.observation_space.spaces.keys()) == ['action_mask', 'obs']
    ho, o = henv.reset(), env.reset()
    for t in range(T):
        assert np.array_equal(np.stack([x['obs'] for x in ho]), np.stack([x['obs'] for x in o]).astype(np.float32))
        assert np.array_equal(np.stack([x['action_mask'] for x in ho]), np.stack([x['action_mask'] for x in o]).astype(np.float32))
        a = np.asarray(env.offline_action)
        ho, hr, hd, hi = henv.step(a)
        o, r, dn, info = env.step(a)
        assert np.allclose(hr, r, rtol=0, atol=0) and hd == dn and len(hi) == B
    assert max(hr) > 0
    henv.close()


def _deep_equal(u, v):
    """Bit-level equality of the nested list / dict / ndarray / scalar objects the reference-shaped API returns, types included."""
    if isinstance(u, dict):
        return isinstance(v, dict) and set(u) == set(v) and all(_deep_equal(u[k], v[k]) for k in u)
    if isinstance(u, (list, tuple)):
        return isinstance(v, (list, tuple)) and len(u) == len(v) and all(_deep_equal(a, b) for a, b in zip(u, v))
    if isinstance(u, np.ndarray):
        return isinstance(v, np.ndarray) and u.dtype == v.dtype and u.shape == v.shape and np.array_equal(u, v, equal_nan=True)
    return type(u) == type(v) and u == v


@pytest.mark.parametrize('kind', ['plain', 'rllib_mask', 'd3rl_mask', 'info_fetch', 'conti', 'conti_mask', 'seq', 'seq_d3rl', 'seq_mask_fetch',
                                  'widedeep', 'one_env'])
def test_reference_shaped_step_is_one_record_and_bit_identical(tmp_path, kind):
    """The reference-shaped modes (lists / ndarrays / dicts out, base.py:256-263, slate.py:244-279) through
    rl4rs_env_step_record - ONE library call, one device record, one copy into pinned memory per transition - against the
    composed path (config['no_fused_step']): every returned object identical in value, dtype, shape and python type over two
    episodes; info['click_p'] under simulator_info_fetch; the host-side offline_action cache; batch-of-one unwrapping."""
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from rl4rs_amd.env.seqslate import SeqSlateRecEnv, SeqSlateState
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=4)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    seq = kind.startswith('seq')
    T = 18 if seq else 9
    recs = synth.make_records(200, pages=2 if seq else 1, seed=3, hash_size=2000, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    B = 1 if kind == 'one_env' else 40
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'),
           "iteminfo_file": os.path.join(d, 'c.csv'), "cache_size": 128, "model_seed": 3}
    if kind in ('rllib_mask', 'conti_mask', 'seq_mask_fetch', 'one_env'):
        cfg['support_rllib_mask'] = True
    if kind in ('d3rl_mask', 'seq_d3rl'):
        cfg['support_d3rl_mask'] = True
    if kind in ('info_fetch', 'seq_mask_fetch'):
        cfg['simulator_info_fetch'] = True
    if kind in ('conti', 'conti_mask'):
        cfg['support_conti_env'] = True
    if kind == 'widedeep':
        cfg['algo'] = 'widedeep'

    def run(fused, copies=False):
        c = dict(cfg, no_fused_step=not fused, copy_outputs=copies)
        if seq:
            env = rl4rs_amd.make('SeqSlateRecEnv-v0', recsim=SeqSlateRecEnv(c, state_cls=SeqSlateState))
        else:
            env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(c, state_cls=SlateState))
        env.seed(11)
        out = []
        for ep in range(2):
            out.append(env.reset())
            if fused and B > 1 and not cfg.get('support_conti_env'):
                # the reset came back through rl4rs_env_observe_record_host: the first logged action is already on the device
                first = env.offline_action
                assert type(first).__name__ == 'OfflineActionList' and first._dev is not None
                # every access hands out a FRESH list like the reference (slate.py:150-161): editing one does not leak into the
                # next read, which still remembers the device copy of the ids
                again = env.offline_action
                assert again is not first and again == first and again._dev is not None
                first[0] = -7
                assert env.offline_action[0] == again[0] != -7
            for t in range(T):
                a = env.offline_action
                obs, reward, done, info = env.step(a)
                if copies and fused:
                    # config['copy_outputs']: pageable arrays that own their memory (not views of the step's pinned block)
                    arrs = [obs] if isinstance(obs, np.ndarray) else ([obs[0]['obs'], obs[0]['action_mask']] if isinstance(obs, list) else
                                                                     [obs['obs'], obs['action_mask']])
                    for arr in arrs:
                        root = arr
                        while isinstance(getattr(root, 'base', None), np.ndarray):
                            root = root.base
                        assert root.flags.owndata and root.nbytes <= max(a_.nbytes for a_ in arrs) * B
                info_copy = [dict((k, np.array(v)) for k, v in i.items()) for i in (info if isinstance(info, list) else [info])]
                out.append((a, obs, reward, done, info_copy, np.asarray(env.samples.last_actions.cpu().numpy())))
            out.append((env.samples.prev_actions, env.samples.get_violation(), env.offline_reward))
        assert (getattr(env.sim, '_stepper', None) is not None) == fused
        return out

    a, b = run(True), run(False)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert _deep_equal(x, y), (kind, i)
    if kind in ('plain', 'rllib_mask', 'd3rl_mask', 'one_env'):
        c = run(True, copies=True)
        for i, (x, y) in enumerate(zip(a, c)):
            assert _deep_equal(x, y), (kind, 'copy_outputs', i)
    steps = [x for x in a if isinstance(x, tuple) and len(x) == 6]
    assert any(np.sum(np.abs(np.asarray(s[2], dtype=np.float64))) > 0 for s in steps)            # rewards were paid
    if 'fetch' in kind:
        assert any('click_p' in s[4][0] and s[4][0]['click_p'].shape == (9,) for s in steps)
    if kind == 'd3rl_mask':
        assert steps[0][1].dtype == np.float64 and steps[0][1].shape == (B, 256 + 9 + 1)


def test_overridden_plugin_methods_are_honoured(tmp_path):
    """state_cls / the env class are the reference's extension points: a subclass that overrides act (or obs_fn / forward /
    _reward_due) must see its override run - such an env leaves the fused entry points alone and composes the transition."""
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=4)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(60, seed=3, hash_size=2000, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    B, T = 16, 9
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'),
           "iteminfo_file": os.path.join(d, 'c.csv'), "cache_size": 32, "model_seed": 3}
    calls = {'act': 0, 'forward': 0}

    class CountingState(SlateState):
        def act(self, actions):
            calls['act'] += 1
            SlateState.act(self, actions)

    class DoubledReward(SlateRecEnv):
        def forward(self, model, samples):
            calls['forward'] += 1
            r = SlateRecEnv.forward(self, model, samples)
            return r * 2 if torch.is_tensor(r) else [2 * x for x in r]

    for tensors in (False, True):
        c = dict(cfg, return_tensors=tensors)
        ref = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(dict(c), state_cls=SlateState))
        e1 = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(dict(c), state_cls=CountingState))
        e2 = rl4rs_amd.make('SlateRecEnv-v0', recsim=DoubledReward(dict(c), state_cls=SlateState))
        for e in (ref, e1, e2):
            e.seed(5)
            e.reset()
        calls.update(act=0, forward=0)
        for t in range(T):
            a = ref.offline_action
            _, r0, _, _ = ref.step(a)
            _, r1, _, _ = e1.step(a)
            _, r2, _, _ = e2.step(a)
        assert calls['act'] == T and calls['forward'] == T
        assert getattr(ref.sim, '_stepper', None) is not None and getattr(e1.sim, '_stepper', None) is None
        r0, r1, r2 = [np.asarray(x.cpu() if torch.is_tensor(x) else x, dtype=np.float64) for x in (r0, r1, r2)]
        assert r0.max() > 0 and np.array_equal(r0, r1) and np.array_equal(2 * r0, r2)


@pytest.mark.parametrize('flag', [None, 'support_rllib_mask', 'support_d3rl_mask'])
def test_configs0_batch_256_replay_against_reference_vectors(flag):
    """BASELINE configs[0] (SlateRecEnv-v0, batch 256, 9-slot, offline_action replay) through the gym facade in the reference's
    own list / ndarray modes, against vectors the REFERENCE's SlateState produced for these 256 records
    (tests/golden/slate256_discrete.npz, make_golden.py): logged actions, the observation-side mask rows of the rllib mode, the
    masked_actions | cur_steps tail of the d3rlpy mode, prev_actions, violation and the logged reward - bit for bit."""
    import rl4rs_amd
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from helpers import load_scenario, GOLDEN
    m, cfg, records, g = load_scenario('slate256_discrete')
    cfg = dict(cfg, sample_file=os.path.join(GOLDEN, m['records']), is_eval=True, cache_size=256, model_seed=3)
    for k in ('support_rllib_mask', 'support_d3rl_mask'):
        cfg.pop(k, None)
    if flag:
        cfg[flag] = True
    B, T = cfg['batch_size'], cfg['max_steps']
    assert B == 256 and T == 9
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    obs = env.reset(reset_file=True)
    if flag == 'support_rllib_mask':
        assert np.array_equal(np.stack([o['action_mask'] for o in obs]), g['obsmask_init'])
    for t in range(T):
        a = env.offline_action
        assert np.array_equal(np.asarray(a), g['offline_action_%d' % t]), t
        obs, reward, done, info = env.step(a)
        if flag == 'support_rllib_mask':
            assert np.array_equal(np.stack([o['action_mask'] for o in obs]), g['obsmask_%d' % t]), t
            assert obs[0]['action_mask'].dtype == np.int64 and obs[0]['obs'].dtype == np.float32
        elif flag == 'support_d3rl_mask':
            assert obs.dtype == np.float64 and obs.shape == (B, 256 + T + 1)
            assert np.array_equal(obs[:, 256:256 + T], g['d3rl_prev_%d' % t].astype(np.float64)), t
            assert np.array_equal(obs[:, -1:], g['d3rl_cur_%d' % t].astype(np.float64)), t
        else:
            assert obs.dtype == np.float32 and obs.shape == (B, 256)
        assert np.array_equal(env.samples.prev_actions, g['prev_actions_%d' % t]), t
        assert np.array_equal(np.asarray(env.offline_reward, dtype=np.float64), g['offline_reward_%d' % t]), t
    assert getattr(env.sim, '_stepper', None) is not None
    assert np.array_equal(env.samples.action_mask, g['action_mask_%d' % (T - 1)])
    assert np.array_equal(env.samples.special_mask, g['special_mask_%d' % (T - 1)])
    assert np.array_equal(env.samples.get_violation(), g['violation_end'])
    viol = g['violation_end']
    r = np.asarray(reward, dtype=np.float64)
    assert (r[viol == 0] == 0).all() and (r[viol == 1] > 0).all()


def test_record_entry_points_validate_and_agree(tmp_path):
    """rl4rs_env_step_record_host / rl4rs_env_observe_record_host at the C ABI: bad arguments come back as error codes with a
    message (never a crash), the record a NULL host block leaves on the device equals what the host-copying form brings home,
    and the int64 mask - copied on the stepper's own stream beside the scorer - is the mask the composed path reports."""
    import ctypes as C
    import torch
    from rl4rs_amd import _lib, device as D
    cfg, records, w = _setup(tmp_path, False, 24, 9, support_rllib_mask=True)
    env = _make(cfg, False)
    env.reset()
    sim, samples = env.sim, env.samples
    _, _, stepper = sim._stepper_for(samples)
    lib = _lib.load()
    want = _lib.STEP_WANT['mask_i64'] | _lib.STEP_WANT['offline_action']
    L, rec = stepper._layout(want, False)
    host = torch.empty(int(L.host_bytes), dtype=torch.uint8, pin_memory=True)
    stream = D._stream()
    acts = torch.zeros(24, dtype=torch.int32, device='cuda')
    # error paths
    assert lib.rl4rs_env_step_record_host(stepper.h, acts.data_ptr(), 0, want, rec.data_ptr(), None, stream) != 0
    assert b'host' in lib.rl4rs_last_error()
    assert lib.rl4rs_env_observe_record_host(stepper.h, 0, want | _lib.STEP_WANT['click_p'], rec.data_ptr(), host.data_ptr(), stream) != 0
    assert b'click_p' in lib.rl4rs_last_error()
    bad = _lib.StepRecord()
    assert lib.rl4rs_stepper_record_layout(stepper.h, 1 << 9, 0, C.byref(bad)) != 0
    assert lib.rl4rs_env_step_record_host(stepper.h, acts.data_ptr(), 7, want, rec.data_ptr(), host.data_ptr(), stream) != 0
    # the failed calls changed nothing: observe twice, once with and once without the host copies
    _lib.check(lib.rl4rs_env_observe_record_host(stepper.h, 0, want, rec.data_ptr(), host.data_ptr(), stream))
    torch.cuda.synchronize()
    first = host.clone()
    _lib.check(lib.rl4rs_env_observe_record_host(stepper.h, 0, want, rec.data_ptr(), None, stream))
    torch.cuda.synchronize()
    dev = rec[:int(L.host_bytes)].cpu()
    lo, hi = int(L.obs), int(L.obs) + 24 * 256 * 4
    assert torch.equal(dev[lo:hi], first[lo:hi])
    mlo = int(L.mask_i64)
    assert torch.equal(dev[mlo:mlo + 24 * 284 * 8], first[mlo:mlo + 24 * 284 * 8])
    mask = first[mlo:mlo + 24 * 284 * 8].numpy().view(np.int64).reshape(24, 284)
    assert np.array_equal(mask, samples._obs_mask())
    obs = first[lo:hi].numpy().view(np.float32).reshape(24, 256)
    assert np.array_equal(obs, np.stack([o['obs'] for o in env.state]))
