"""BASELINE.json full size (SlateRecEnv-v0, B=4096, 284 items, 9 slots, hash 100000): integer state bit-exact vs the
numpy oracle on the whole batch; observations / rewards of NPICK = 512 envs vs the fp64 DIEN oracle (oracle/dien_torch.py in
float64 on the host's threads; rows are independent, so a subset is a valid check of the full-size launch geometry - 512 of
4096 envs cover every workgroup position class of the 32- and 64-row kernels); duplicate-record invariance; and the BENCH's
own configuration (train-mode sampling with duplicates, history dedup, row-order hint, fused step)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NPICK = 512


def _fp64_scorer(cfg, seed=7):
    """The fp64 checker: same restatement as oracle/dien.py (tests/test_oracle_dien.py pins the two together to 1e-11), on
    row-parallel host threads so that hundreds of envs per step stay within seconds."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from oracle.dien_torch import TorchDien
    return TorchDien(init_dien_weights(cfg, seed=seed), cfg, workers=max(1, min(32, (os.cpu_count() or 2) // 2)), dtype=torch.float64)


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
    orc = _fp64_scorer(cfg)
    pick = np.sort(np.random.RandomState(0).choice(B, NPICK, replace=False))

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


def _seqslate_full_episode(tmp_path, T, every_first_page_step=True):
    """SeqSlateRecEnv-v0 at B=4096: one offline_action episode with the integer state of the WHOLE batch bit-exact vs the oracle
    at every step, observations and page rewards of a 512-env subset vs the fp64 DIEN oracle (at every page boundary, the
    step before it and the end; on the first page at every step when asked).  Returns (env, number of reward steps)."""
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.env.seqslate import SeqSlateRecEnv, SeqSlateState
    from oracle.state import OracleState
    from oracle.env import reward_from_probs, is_reward_step
    B = 4096
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=1234)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(B, pages=4, seed=1001, illegal_frac=0.05, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    cfg = _full_cfg(d, B, T, recs, support_rllib_mask=True)
    env = rl4rs_amd.make('SeqSlateRecEnv-v0', recsim=SeqSlateRecEnv(cfg, state_cls=SeqSlateState))
    obs = env.reset(reset_file=True)
    st = OracleState(cfg, recs, seq=True)
    orc = _fp64_scorer(cfg)
    pick = np.sort(np.random.RandomState(1).choice(B, NPICK, replace=False))
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
        # observations against the fp64 scorer at every page boundary, the step before it and the end (each costs NPICK fp64
        # DIEN rows on the host); masks at EVERY step, the whole integer state at every page boundary
        check_obs(obs, (every_first_page_step and t < 10) or t % 9 in (0, 8) or t == T - 1)
        if t % 9 == 8 or t == T - 1:
            assert np.array_equal(env.samples.prev_actions, st.prev_actions)
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
    assert np.array_equal(env.samples.get_violation(), st.get_violation())
    return env, n_reward_steps


def test_full_size_seqslate_t32_episode_then_ppo_iteration(tmp_path):
    """BASELINE configs[2]: SeqSlateRecEnv-v0, B=4096, 32-step horizon (pages of 9: rewards after steps 9, 18, 27; the last 5
    steps never pay, seqslate.py:138).  Integer state of the WHOLE batch bit-exact vs the oracle, observations
    and page rewards of a 512-env subset vs the fp64 DIEN oracle, then one PPO train call of the on-device loop on the same env
    (script/modelfree_train.py:42-44,179-247)."""
    import torch
    from rl4rs_amd.train import Trainer
    B, T = 4096, 32
    env, n_reward_steps = _seqslate_full_episode(tmp_path, T)
    assert n_reward_steps == 3
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


@pytest.mark.parametrize('T', [36, 32])
def test_full_size_seqslate_a2c(tmp_path, T):
    """BASELINE configs[3]'s per-GPU shard at full size: SeqSlateRecEnv-v0, B=4096, the parity horizon T=36 (4 pages, 4 reward
    steps: script/modelfree_train.py:43, simulator_eval.py:20) and the bench horizon T=32, followed by A2C train calls of the
    on-device loop (modelfree_train.py:248-304: one summed-loss gradient over the whole 147 456 / 131 072-sample rollout,
    global-norm clip 10, Adam 1e-4) tracked against the float64 restatement (oracle/policy.py a2c_train_call), teacher-forced
    on the device's own rollouts."""
    import torch
    from rl4rs_amd.train import Trainer
    from oracle import policy as OP
    B = 4096
    env, n_reward_steps = _seqslate_full_episode(tmp_path, T, every_first_page_step=False)
    assert n_reward_steps == T // 9
    tr = Trainer(env, algo='A2C', seed=11, init_seed=3, keep_last_batch=True)
    assert tr.R == 1 and tr.buf['obs'].shape[0] == B * T
    flat = tr.params().cpu().numpy().astype(np.float64)
    state = (flat, np.zeros_like(flat), np.zeros_like(flat), 0)

    def unpack(bits):
        b = bits.view(np.uint32)
        return ((b[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(b.shape[0], -1)[:, :284].astype(np.float64)

    for it in range(2):
        out = tr.train_iteration()
        lb = tr.last_batch
        batch = dict((k, lb[k].cpu().numpy()) for k in ('obs', 'act', 'mask', 'adv', 'ret'))
        # the rollout itself: every sampled action legal for its slot, a reward only at the page boundaries
        acts = batch['act'].reshape(T, B)
        assert (acts >= 1).all() and (acts < 284).all()
        rew = tr.buf['rew'].view(T, B).cpu().numpy()
        pay = np.array([(t + 1) % 9 == 0 for t in range(T)])
        assert (rew[~pay] == 0).all() and (rew[pay] > 0).any()
        assert env.samples.get_violation().all()             # the masked policy never violates the slate rules
        # returns with gamma = 1: ret[t] = sum of the rewards from t on
        ret = batch['ret'].reshape(T, B)
        assert np.allclose(ret, np.cumsum(rew[::-1], axis=0)[::-1], rtol=1e-6, atol=1e-4)
        state, sums, norm, g = OP.a2c_train_call(state, batch, 1e-4, unpack)
        got = tr.params().cpu().numpy()
        assert norm > 10.0                                    # summed losses over 1e5 samples: the clip is active
        # the gradient itself (before the clip) against float64 autograd over the whole rollout batch
        g_ref = g * max(norm / 10.0, 1.0)
        g_dev = tr.grad.cpu().numpy().astype(np.float64)
        assert np.abs(g_dev - g_ref).max() < 2e-3 * np.abs(g_ref).max(), (it, np.abs(g_dev - g_ref).max(), np.abs(g_ref).max())
        # Adam's first steps move a weight by ~lr * sign(g): an entry whose gradient is a rounding-level residue of the 1e5-term
        # sums may step the other way in fp32; every entry whose fp32 gradient is good to 1 % must land on the float64 weight
        solid = np.abs(g_dev - g_ref) <= 0.01 * np.abs(g_ref)
        assert solid.mean() > 0.9, solid.mean()
        assert np.abs(got - state[0])[solid].max() < 5e-6, (it, np.abs(got - state[0])[solid].max())
        assert np.abs(got - state[0]).max() <= 2.2e-4 * (it + 1)
        N = B * T
        assert np.allclose([out['policy_loss'] / N, out['vf_loss'] / N, out['entropy'] / N], sums[:3] / N, rtol=2e-3, atol=1e-4), (out, sums)
        assert out['episode_reward_mean'] > 0
    tr.close()


def test_full_size_continuous_action_episode(tmp_path):
    """BASELINE configs[4], its 1-GPU half: SlateRecEnv-v0 with continuous 32-d actions resolved by the masked float64 K-NN,
    B=4096.  Chosen items / integer state bit-exact vs the oracle on the whole batch (actions = logged items' embeddings
    plus noise, so the K-NN and its masks decide), observations and rewards of a 512-env subset vs the fp64 DIEN oracle."""
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
    orc = _fp64_scorer(cfg)
    pick = np.sort(np.random.RandomState(2).choice(B, NPICK, replace=False))
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


def test_the_bench_configuration_itself(tmp_path):
    """The path the headline number runs: bench.py's own config and env construction - train-mode sampling
    (np.random.choice of 4096 envs from a 2048-line cache window, base.py:92-100: ~57 % of the envs share a user history with
    another env), history de-duplication + the env -> slot table, the row-order hint (rl4rs_dien_set_row_order), zero-copy
    tensors, the fused step entry point, the automatic 32- / 64-row AUGRU forms - over TWO episode-batches on the same handles.
    Integer state of all 4096 envs bit-exact against the oracle replaying the very records the env sampled; observations of
    512 envs at every step and their rewards against the fp64 scorer; envs that drew the same log line carry bit-identical
    observation rows and rewards."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle.state import OracleState
    from oracle.env import reward_from_probs

    class Args(object):
        env, batch, horizon, log_records, scorer, algo, conti = 'slate', 4096, 9, 8193, 'auto', 'dien', False

    cfg, _ = bench.make_config(Args(), str(tmp_path), 0)
    assert cfg['is_eval'] is False and cfg['cache_size'] == 2048 and cfg['return_tensors'] and 'no_row_order' not in cfg
    B, T = cfg['batch_size'], cfg['max_steps']
    env = bench.build_env(cfg, False)
    env.seed(1000)
    env.sim._recData.store.preload(torch.device('cuda', torch.cuda.current_device()))
    orc = _fp64_scorer(cfg, seed=cfg['model_seed'])
    rs = np.random.RandomState(3)
    for episode in range(2):
        obs = env.reset()
        recs = list(env.samples.records)                      # what RecDataBase.sample drew for this batch
        rows = np.asarray(env.samples.records.rows)
        uniq, first, inv, counts = np.unique(rows, return_index=True, return_inverse=True, return_counts=True)
        assert 1500 < len(uniq) < 2048 and env.samples._hist_unique[0].shape[0] == len(uniq)
        assert getattr(env.samples, '_row_order', None) is not None
        st = OracleState(cfg, recs)
        # picked envs: half of them from histories that several envs share, the rest at random
        shared = np.flatnonzero(counts[inv] > 1)
        pick = np.unique(np.concatenate([rs.choice(shared, NPICK // 2, replace=False), rs.choice(B, NPICK // 2, replace=False)]))
        pick_t = torch.from_numpy(pick).cuda()
        # one representative per log line, to compare every duplicate env with
        rep_t = torch.from_numpy(first[inv]).cuda()

        def check_obs(o, t):
            seq, dense, cat = st.features()
            ref = orc.obs(seq[pick], dense[pick], cat[pick])
            assert np.abs(o[pick_t].cpu().numpy() - ref).max() < 5e-5, (episode, t)
            assert torch.equal(o, o[rep_t]), (episode, t)                  # same log line + same actions -> same bits

        check_obs(obs, -1)
        for t in range(T):
            a = env.offline_action
            assert np.array_equal(a.cpu().numpy(), np.asarray(st.offline_action))
            obs, reward, done, info = env.step(a)
            st.act(a.cpu().numpy())
            check_obs(obs, t)
        assert getattr(env.sim, '_stepper', None) is not None              # the fused entry point ran
        assert np.array_equal(env.samples.prev_actions, st.prev_actions)
        assert np.array_equal(env.samples.action_mask, st.action_mask)
        assert np.array_equal(env.samples.special_mask, st.special_mask)
        assert np.array_equal(env.samples.get_violation(), st.get_violation())
        assert np.array_equal(np.asarray(env.offline_reward.cpu()), np.asarray(st.offline_reward))
        cs, cd, cc = st.complete_features()
        crow = (pick[:, None] * 9 + np.arange(9)[None, :]).reshape(-1)
        probs = np.zeros((B, 9), dtype=np.float64)
        probs[pick] = orc.prob(cs[crow], cd[crow], cc[crow]).reshape(-1, 9)
        ref = np.asarray(reward_from_probs(st, probs.astype(np.float32)))
        r = reward.cpu().numpy()
        assert np.allclose(r[pick], ref[pick], rtol=1e-5, atol=1e-5), episode
        assert np.array_equal(r, r[first[inv]])
        viol = st.get_violation()
        assert (r[viol == 0] == 0).all() and (r[viol == 1] > 0).all()
        env.sim.model.device_net.check_status()


def test_four_times_the_baseline_batch_equals_its_chunks(tmp_path):
    """Beyond BASELINE.json's size (one GPU has the memory for far bigger shards): B = 16 384 envs with 16 384 DISTINCT user
    histories - 64-row AUGRU workgroups for the observation launches too, a 3.2 GB sequence cache per input, 147 456 complete
    rows in the reward forward - against the same records run as four batches of 4096: every observation of every step and
    every reward bit-identical (rows are independent and the kernels batch-position invariant, so this is a size-independent
    check of the big launch geometry; the 4096-env batches are the ones the tests above pin to the oracle).  One step further,
    32 768 distinct histories, the per-input cache would pass the 4 GB a 32-bit buffer offset can address: refused when the
    scorer handle is created, loudly - more envs than that per GPU are several env handles."""
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth, _lib
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    BIG, CH, T = 16384, 4096, 9
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=1234)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(2 * BIG, seed=1000, illegal_frac=0.05, special_ids=synth.special_ids_from_text(text))

    def episode(rs):
        synth.write_records(os.path.join(d, 'log.csv'), rs)
        env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(_full_cfg(d, len(rs), T, rs), state_cls=SlateState))
        out = [env.reset(reset_file=True).clone()]
        for t in range(T):
            obs, reward, done, info = env.step(env.offline_action)
            out.append(obs.clone())
        res = torch.stack(out).cpu(), reward.clone().cpu()
        del env
        torch.cuda.empty_cache()
        return res

    big_obs, big_r = episode(recs[:BIG])
    assert torch.isfinite(big_obs).all() and torch.isfinite(big_r).all() and float(big_r.abs().max()) > 0
    for k in range(0, BIG, CH):
        o, r = episode(recs[k:k + CH])
        assert torch.equal(o, big_obs[:, k:k + CH]), k
        assert torch.equal(r, big_r[k:k + CH]), k
    with pytest.raises(_lib.Rl4rsHipError, match='4 GB'):
        episode(recs)
