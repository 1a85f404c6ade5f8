"""Supervised training of the simulator families on the device (rl4rs_simtrain_* for dnn / widedeep / lstm, rl4rs_dientrain_* for dien) against torch float64 autograd of the
numpy-restated model (oracle/simnets.py): gradients with and without dropout, the Adam update, and a short run that must
fit a learnable labelling.  Gradient tolerance: 2e-4 of the largest gradient entry (fp32 kernels vs fp64)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = {"maxlen": 64, "class_num": 2, "dense_feature_num": 432, "category_feature_num": 21, "category_hash_size": 3000,
       "seq_num": 2, "emb_size": 128, "hidden_units": 128}


def _batch(N, rs):
    dense = np.abs(rs.randn(N, 432)).astype(np.float32)
    cat = rs.randint(0, 3000, size=(N, 21)).astype(np.int32)
    labels = rs.randint(0, 2, size=N).astype(np.int32)
    return dense, cat, labels


@pytest.fixture
def recur_rows():
    """pins the row-tile form of the persistent training recurrences for one test (rl4rs_recur_train_set_rows), automatic afterwards"""
    from rl4rs_amd import _lib

    def pin(rows):
        _lib.check(_lib.load().rl4rs_recur_train_set_rows(rows))
    yield pin
    pin(0)


@pytest.mark.parametrize('algo', ['dnn', 'widedeep', 'lstm'])
@pytest.mark.parametrize('rate', [0.0, 0.2])
@pytest.mark.parametrize('N', [256, 700])
def test_gradients_match_autograd(algo, rate, N):
    _check_simnet_gradients(algo, rate, N)


@pytest.mark.parametrize('rows', [8, 32])
@pytest.mark.parametrize('N', [256, 203])
def test_lstm_gradients_in_both_recurrence_tile_forms(recur_rows, rows, N):
    """the keras GRUs of the lstm family through the 8-row (v_mfma_f32_4x4x1, recur8.hpp) and the 32-row persistent recurrences,
    each against float64 autograd at the same bar; 203 rows: ragged last tile of both forms"""
    from rl4rs_amd import _lib
    recur_rows(rows)
    _check_simnet_gradients('lstm', 0.2, N)
    with pytest.raises(_lib.Rl4rsHipError):
        recur_rows(16)


def _check_simnet_gradients(algo, rate, N):
    import torch
    from rl4rs_amd.nets.simnets import init_simnet_weights
    from rl4rs_amd.device import DeviceSimTrainer
    from oracle.simnets import loss_and_grad
    rs = np.random.RandomState(N + int(rate * 10))
    w = init_simnet_weights(CFG, algo, seed=3, emb_scale=0.5, bias_noise=0.2)
    dense, cat, labels = _batch(N, rs)
    seqs = [rs.randint(0, 284, size=(N, 64)).astype(np.int32) for _ in range(2)]
    seqs[1][::2] = 0
    tr = DeviceSimTrainer(CFG, w, max_batch=N, algo=algo)
    t = lambda a: torch.from_numpy(a).cuda()
    dseqs = [t(q) for q in seqs] if algo != 'dnn' else None
    loss = tr.grad(t(dense), t(cat), t(labels), dseqs, dropout_rate=rate, seed=5, step=2)
    g = dict((k, v.cpu().numpy()) for k, v in tr.gradients().items())
    m1 = m2 = None
    if rate > 0:
        # the counter RNG is a pure function of (seed, step, row, column): a second call redraws the same masks
        loss2 = tr.grad(t(dense), t(cat), t(labels), dseqs, dropout_rate=rate, seed=5, step=2)
        assert torch.equal(loss, loss2)
        m1, m2 = [m.cpu().numpy().astype(np.float64) for m in tr.masks(N)]
        assert not np.array_equal(m1, m2)
        keep = np.mean(m1)
        assert abs(keep - (1 - rate)) < 0.02
    loss_ref, g_ref = loss_and_grad(algo, w, dense, cat, labels, seqs, m1, m2, rate)
    assert abs(float(loss.item()) - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    assert set(g) == set(g_ref)
    for k in g_ref:
        scale = np.abs(g_ref[k]).max()
        assert np.abs(g[k] - g_ref[k]).max() < 2e-4 * max(scale, 1e-8), (k, np.abs(g[k] - g_ref[k]).max(), scale)
    tr.close()


def test_adam_step_and_training_fits(tmp_path):
    import torch
    from rl4rs_amd.nets.simnets import init_simnet_weights
    from rl4rs_amd.device import DeviceSimTrainer, DeviceSimnet
    rs = np.random.RandomState(0)
    w = init_simnet_weights(CFG, 'dnn', seed=1, emb_scale=0.05)
    N = 256
    dense, cat, _ = _batch(N, rs)
    labels = (dense[:, :8].sum(axis=1) > np.median(dense[:, :8].sum(axis=1))).astype(np.int32)     # learnable rule
    tr = DeviceSimTrainer(CFG, w, max_batch=N)
    t = lambda a: torch.from_numpy(a).cuda()
    # first step: keras Adam with m = v = 0 moves every touched weight by lr * g / (|g| + eps * sqrt(1-b2)) ~ lr * sign(g)
    before = dict((k, v.clone()) for k, v in tr.weights().items())
    tr.grad(t(dense), t(cat), t(labels), dropout_rate=0.0)
    g = tr.gradients()
    tr.iteration = 0
    tr.step(t(dense), t(cat), t(labels), lr=1e-3, dropout_rate=0.0)
    after = tr.weights()
    for k in ('out_w', 'fc_w', 'dense_w1'):
        gk = g[k]
        lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
        expect = before[k] - lr_t * (0.1 * gk) / (torch.sqrt(0.001 * gk * gk) + 1e-7)
        assert torch.allclose(after[k], expect, rtol=0, atol=2e-6), k
    untouched = np.setdiff1d(np.arange(3000), np.unique(cat))
    assert torch.equal(after['cat_emb'][untouched], before['cat_emb'][untouched])      # zero gradient rows do not move
    losses = []
    for it in range(200):
        losses.append(float(tr.step(t(dense), t(cat), t(labels), lr=1e-3, dropout_rate=0.2, seed=9).item()))
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    # the trained weights drop into the inference scorer
    trained = dict((k, v.cpu().numpy()) for k, v in tr.weights().items())
    net = DeviceSimnet(CFG, trained, max_rows=N, max_slots=1, algo='dnn')
    slots = torch.zeros((2, N), dtype=torch.int32).cuda()
    _, prob = net.forward(N, 1, t(dense), t(cat), slots, want_obs=False, want_prob=True)
    acc = ((prob.cpu().numpy() > 0.5).astype(np.int32) == labels).mean()
    assert acc > 0.8, acc
    net.close()
    tr.close()


@pytest.mark.parametrize('algo', ['dnn', 'widedeep', 'lstm', 'dien'])
def test_training_set_from_logs_and_fit(tmp_path, algo):
    """SimulatorTrainer: the device-built training set equals the reference's construction (data_preprocess.py:91-131:
    category = user_cat + [sequence_id] + exposed + [item_j], dense = user_dense + item vectors of the page + item_j,
    label = user_feedback[j]); a short fit lowers the loss and the trained weights drop into the env."""
    import os
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.data import CatalogTables
    from rl4rs_amd.simtrain import SimulatorTrainer
    from rl4rs.env.slate import SlateRecEnv, SlateState
    B = 64
    d = str(tmp_path)
    cat_path, log_path = os.path.join(d, 'item_info.csv'), os.path.join(d, 'log.csv')
    cat_text = synth.make_catalog_text(seed=21)
    synth.write_text(cat_path, cat_text)
    records = synth.make_records(B, pages=1, seed=8, illegal_frac=0.0, hash_size=5000,
                                 special_ids=synth.special_ids_from_text(cat_text))
    synth.write_records(log_path, records)
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 5000, "seq_num": 2, "emb_size": 128,
           "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32,
           "sample_file": log_path, "iteminfo_file": cat_path, "is_eval": True, "cache_size": B, "model_seed": 3,
           "algo": algo, "return_tensors": True}
    sim = SlateRecEnv(cfg, state_cls=SlateState)
    tr = SimulatorTrainer(sim, minibatch=64, seed=1)
    dense, cat, labels, seqs = tr.dataset_from_logs()
    assert len(seqs) == 2 and seqs[0].shape == (B * 9, 64) and int(seqs[1].abs().sum()) == 0
    assert dense.shape == (B * 9, 432) and cat.shape == (B * 9, 21) and labels.shape == (B * 9,)
    tab = CatalogTables(cat_path, 284, 32)
    dn, cn, ln = dense.cpu().numpy(), cat.cpu().numpy(), labels.cpu().numpy()
    for b in (0, 17, B - 1):
        f = records[b].split('@')
        exposed = [int(x) for x in f[3].split(',')][:9]
        feedback = [int(x) for x in f[4].split(',')][:9]
        portrait = [float(x) for x in f[6].split(',')]
        for j in (0, 4, 8):
            row = b * 9 + j
            exp_cat = [int(x) for x in portrait[:10]] + [1] + exposed + [exposed[j]]
            assert list(cn[row]) == exp_cat
            exp_dense = np.concatenate([np.asarray(portrait[10:], dtype=np.float32)] + [tab.item_vec[i] for i in exposed] +
                                       [tab.item_vec[exposed[j]]])
            assert np.array_equal(dn[row], exp_dense)
            assert ln[row] == feedback[j]
    losses = tr.fit(windows=1, epochs=12)
    assert len(losses) == 12 * (B * 9 // 64)
    assert np.mean(losses[-9:]) < 0.8 * np.mean(losses[:9]), (np.mean(losses[:9]), np.mean(losses[-9:]))
    tr.install()
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=sim)
    env.reset(reset_file=True)
    total = 0.0
    for _ in range(9):
        obs, reward, done, info = env.step(env.offline_action)
        total += float(reward.sum())
    assert np.isfinite(total) and total > 0


DIEN_CFG = {"maxlen": 64, "class_num": 2, "dense_feature_num": 432, "category_feature_num": 21, "category_hash_size": 3000,
            "seq_num": 2, "emb_size": 128, "hidden_units": 128}


@pytest.mark.parametrize('rate', [0.0, 0.2])
def test_dien_gradients_match_autograd(rate):
    """rl4rs_dientrain_grad: every parameter gradient of the DIEN training step (head, category self-attention, dense tower
    with dropout, attention MLP, AUGRU and first-GRU BPTT, both embedding tables) against torch float64 autograd."""
    _check_dien_gradients(rate, 40)


@pytest.mark.parametrize('rows', [4, 8, 32])
@pytest.mark.parametrize('N', [40, 43, 72])
def test_dien_gradients_in_both_recurrence_tile_forms(recur_rows, rows, N):
    """the first GRU (Hd = 128) and the AUGRU (Hd = 256) with their BPTT through the 8-row and the 32-row persistent kernels
    (rows = 4: the AUGRU in 4-row workgroups - one 4-row tile per wave - and the first GRU, whose width has no such form, in 8-row ones;
    N = 43: ragged last tile of every form);
    N = 72: 4 608 (row, step) pairs - the sample-axis reductions of the recurrent layers' weight gradients ([256 x 512], [512 x 64],
    ...) take the LDS-tiled 128 x 128 form (k_gemm_tn_t128, from 4 096 samples) with its chunk sum"""
    recur_rows(rows)
    _check_dien_gradients(0.2, N)


def _check_dien_gradients(rate, N):
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDienTrainer
    from oracle.dien import loss_and_grad
    rs = np.random.RandomState(7)
    w = init_dien_weights(DIEN_CFG, seed=3, emb_scale=0.5, bias_noise=0.2)
    dense, cat, labels = _batch(N, rs)
    cat[:, 10:] = rs.randint(0, 284, size=(N, 11))
    seq = rs.randint(0, 284, size=(N, 2, 64)).astype(np.int32)
    seq[: N // 3, 0, :20] = 0
    seq[::2, 1, :] = 0
    tr = DeviceDienTrainer(DIEN_CFG, w, max_batch=N)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dseqs = [t(seq[:, i]) for i in range(2)]
    loss = tr.grad(t(dense), t(cat), t(labels), dseqs, dropout_rate=rate, seed=5, step=2)
    g = dict((k, v.cpu().numpy()) for k, v in tr.gradients().items())
    m1 = m2 = None
    if rate > 0:
        m1, m2 = [m.cpu().numpy().astype(np.float64) for m in tr.masks(N)]
    loss_ref, g_ref = loss_and_grad(w, DIEN_CFG, seq, dense, cat, labels, m1, m2, rate)
    assert abs(float(loss.item()) - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    assert set(g) == set(g_ref)
    for k in sorted(g_ref):
        scale = np.abs(g_ref[k]).max()
        err = np.abs(g[k] - g_ref[k]).max()
        assert err < 5e-4 * max(scale, 1e-8) + 1e-9, (k, err, scale)
    tr.close()


def test_fit_from_tfrecord(tmp_path):
    """The reference's training-set format end to end: samples -> FeatureUtil.to_tfrecord -> SimulatorTrainer.fit_tfrecord."""
    import os
    from rl4rs_amd import synth
    from rl4rs_amd.simtrain import SimulatorTrainer
    from rl4rs_amd.utils.datautil import FeatureUtil
    from rl4rs.env.slate import SlateRecEnv, SlateState
    B = 32
    d = str(tmp_path)
    cat_path, log_path = os.path.join(d, 'item_info.csv'), os.path.join(d, 'log.csv')
    cat_text = synth.make_catalog_text(seed=21)
    synth.write_text(cat_path, cat_text)
    synth.write_records(log_path, synth.make_records(B, pages=1, seed=8, illegal_frac=0.0, hash_size=5000,
                                                     special_ids=synth.special_ids_from_text(cat_text)))
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 5000, "seq_num": 2, "emb_size": 128,
           "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32,
           "sample_file": log_path, "iteminfo_file": cat_path, "is_eval": True, "cache_size": B, "model_seed": 3,
           "algo": "widedeep", "return_tensors": True}
    sim = SlateRecEnv(cfg, state_cls=SlateState)
    tr = SimulatorTrainer(sim, minibatch=32, seed=2)
    # the device-built training set of one window, written in the reference's TFRecord layout
    dense, cat, labels, seqs = [x.cpu().numpy() if hasattr(x, 'cpu') else [y.cpu().numpy() for y in x] for x in tr.dataset_from_logs()]
    data = [[0, [seqs[0][i].tolist(), seqs[1][i].tolist()], dense[i].tolist(), cat[i].tolist(), [0] * 9, int(labels[i])]
            for i in range(len(labels))]
    path = os.path.join(d, 'train.tfrecord')
    FeatureUtil(cfg).to_tfrecord(data, path)
    losses = tr.fit_tfrecord(path, steps=60)
    assert len(losses) == 60 and np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < 0.9 * np.mean(losses[:10]), (np.mean(losses[:10]), np.mean(losses[-10:]))
