"""GPU parity of the HIP DIEN scorer against the numpy oracle restatement (fp64) with seeded synthetic
weights.  Tolerances (fp32-accumulating kernels vs fp64 oracle): obs 5e-5 abs (activations are O(1..10)),
probabilities 5e-6 abs; the fp32 oracle itself differs from the fp64 one by ~1e-5 / 1e-6.  Every test runs in both
scorer modes (exact fp32 MFMA and the fp16x2 operand split) against the SAME tolerances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = {"maxlen": 64, "batch_size": 8, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
       "category_feature_num": 21, "category_hash_size": 3000, "seq_num": 2, "emb_size": 128,
       "page_items": 9, "hidden_units": 128, "max_steps": 9, "action_emb_size": 32}


@pytest.fixture(autouse=True, params=['fp32', 'fp16x2'])
def scorer_precision(request):
    CFG['scorer_precision'] = request.param
    yield request.param
    CFG.pop('scorer_precision', None)


def _inputs(R, rs, hash_size):
    seq = rs.randint(0, 284, size=(R, 2, 64)).astype(np.int32)
    seq[: R // 3, 0, :20] = 0            # left padding like short histories
    seq[::2, 1, :] = 0                   # the constant all-zero second sequence of SlateState
    dense = np.abs(rs.randn(R, 432) * 3).astype(np.float32)
    cat = rs.randint(0, hash_size, size=(R, 21)).astype(np.int32)
    cat[:, 10:] = rs.randint(0, 284, size=(R, 11))
    return seq, dense, cat


@pytest.mark.parametrize('R', [5, 64, 333])
def test_dien_rowwise_matches_oracle(R):
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien, DIEN_ALL_FEATURE, DIEN_SCORES, DIEN_QUERY, DIEN_H1
    from oracle.dien import OracleDien
    w = init_dien_weights(CFG, seed=3, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(R)
    seq, dense, cat = _inputs(R, rs, CFG['category_hash_size'])
    net = DeviceDien(CFG, w, max_rows=R, max_slots=R)
    assert net.scorer_mode == CFG['scorer_precision']
    for s in range(2):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    slots = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
    obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots,
                            want_obs=True, want_prob=True)
    torch.cuda.synchronize()
    orc = OracleDien(w, CFG, np.float64)
    allf, parts = orc.features(seq, dense, cat, return_parts=True)
    h1 = net.snapshot(DIEN_H1, R).cpu().numpy()[:R]
    assert np.abs(h1 - parts['h1_0']).max() < 2e-6
    q = net.snapshot(DIEN_QUERY, R).cpu().numpy()[:R]
    assert np.abs(q - parts['query']).max() < 1e-6
    sc = net.snapshot(DIEN_SCORES, R).cpu().numpy()
    assert np.abs(sc[0, :R] - parts['score_0']).max() < 2e-6
    assert np.abs(sc[1, :R] - parts['score_1']).max() < 2e-6
    af = net.snapshot(DIEN_ALL_FEATURE, R).cpu().numpy()[:R]
    assert np.abs(af[:, :256] - parts['h2_0']).max() < 5e-6
    assert np.abs(af[:, 256:512] - parts['h2_1']).max() < 5e-6
    assert np.abs(af[:, 512:640] - parts['dense_feat']).max() < 2e-5 * max(1.0, np.abs(parts['dense_feat']).max())
    # pooled self-attention part of the category branch (the Flatten(category_emb) slice is folded into per-slot
    # head tables at load time and never materialised)
    assert np.abs(af[:, 640:768] - parts['cat_feat'][:, :128]).max() < 2e-6
    obs_ref = orc.obs(seq, dense, cat)
    prob_ref = orc.reward_probs(seq, dense, cat)[:, 1]
    assert np.abs(obs.cpu().numpy() - obs_ref).max() < 5e-5
    assert np.abs(prob.cpu().numpy() - prob_ref).max() < 5e-6
    net.close()


def test_dien_grouped_slots_and_obs_only():
    """Rows of one env share cache slots (group = 9 reward rows); also obs-only / prob-only calls."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from oracle.dien import OracleDien
    w = init_dien_weights(CFG, seed=11, emb_scale=0.5, bias_noise=0.2)
    B, G = 7, 9
    R = B * G
    rs = np.random.RandomState(1)
    seq_env, _, _ = _inputs(B, rs, CFG['category_hash_size'])
    _, dense, cat = _inputs(R, rs, CFG['category_hash_size'])
    net = DeviceDien(CFG, w, max_rows=R + 3, max_slots=B + 2)
    # sequence 0 in slots 2..B+1, sequence 1 shares ONE slot (all-zero ids) like SlateState
    net.encode(0, torch.from_numpy(np.ascontiguousarray(seq_env[:, 0])).cuda(), 2)
    net.encode(1, torch.zeros((1, 64), dtype=torch.int32).cuda(), 1)
    slots = torch.stack([torch.arange(2, B + 2, dtype=torch.int32),
                         torch.full((B,), 1, dtype=torch.int32)]).contiguous().cuda()
    obs, prob = net.forward(R, G, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots,
                            want_obs=True, want_prob=True)
    _, prob2 = net.forward(R, G, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots,
                           want_obs=False, want_prob=True)
    seq_rows = np.repeat(seq_env, G, axis=0).copy()
    seq_rows[:, 1, :] = 0
    orc = OracleDien(w, CFG, np.float64)
    assert np.abs(obs.cpu().numpy() - orc.obs(seq_rows, dense, cat)).max() < 5e-5
    pr = orc.reward_probs(seq_rows, dense, cat)[:, 1]
    assert np.abs(prob.cpu().numpy() - pr).max() < 5e-6
    assert np.array_equal(prob.cpu().numpy(), prob2.cpu().numpy())   # deterministic
    net.close()


def test_dien_bad_arguments():
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from rl4rs_amd._lib import Rl4rsHipError
    import torch
    w = init_dien_weights(CFG, seed=3)
    net = DeviceDien(CFG, w, max_rows=8, max_slots=4)
    with pytest.raises(Rl4rsHipError):
        net.encode(0, torch.zeros((5, 64), dtype=torch.int32).cuda(), 0)      # exceeds max_slots
    slots = torch.zeros((2, 16), dtype=torch.int32).cuda()
    with pytest.raises(Rl4rsHipError):
        net.forward(16, 1, torch.zeros((16, 432)).cuda(), torch.zeros((16, 21), dtype=torch.int32).cuda(), slots)
    bad = dict(CFG, emb_size=64)
    with pytest.raises((Rl4rsHipError, KeyError, ValueError)):
        DeviceDien(bad, w, max_rows=8, max_slots=4)


def test_dien_is_batch_position_invariant():
    """Size-independent property: a row's result does not depend on where it sits in the batch, on the group size,
    or on the run (every kernel is deterministic and row-local; the library is built with -ffp-contract=off so all
    accumulator elements follow one operation sequence)."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    B, G = 10, 8
    w = init_dien_weights(CFG, seed=3, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(0)
    seq, dense, cat = _inputs(B, rs, CFG['category_hash_size'])
    net = DeviceDien(CFG, w, max_rows=B * G, max_slots=B + 1)
    net.encode(0, torch.from_numpy(np.ascontiguousarray(seq[:, 0])).cuda(), 0)
    net.encode(1, torch.zeros((1, 64), dtype=torch.int32).cuda(), B)
    sl = torch.full((2, B), B, dtype=torch.int32).cuda()
    sl[0] = torch.arange(B, dtype=torch.int32).cuda()
    sl = sl.contiguous()
    d1, c1 = torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda()
    obs1, p1 = net.forward(B, 1, d1, c1, sl, True, True)
    obs1b, p1b = net.forward(B, 1, d1, c1, sl, True, True)
    assert torch.equal(obs1, obs1b) and torch.equal(p1, p1b)                       # run to run
    dG, cG = d1.repeat_interleave(G, dim=0).contiguous(), c1.repeat_interleave(G, dim=0).contiguous()
    obsG, pG = net.forward(B * G, G, dG, cG, sl, True, True)
    assert torch.equal(obsG.reshape(B, G, 256), obs1[:, None, :].expand(B, G, 256))  # position / group size
    assert torch.equal(pG.reshape(B, G), p1[:, None].expand(B, G))
    net.close()


@pytest.mark.parametrize('hidden', [128, 96])
def test_chained_dense_tower_is_bit_identical(hidden):
    """fp16x2 mode runs both dense-tower layers in ONE launch (the hidden tile stays in LDS, k_gemm_h16 chain) - same values,
    same k-blocks, same MFMA sequence as two launches with the intermediate in HBM: the whole forward must be bit-identical
    to a handle created with scorer_kernels='no_dense_chain'.  The head with its table half folded into k_cat_attn (vs the
    separate k_head_finish pass, 'no_head_fused') sums in a different order: equal within fp32 rounding only."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    cfg = dict(CFG, hidden_units=hidden)
    B = 70
    w = init_dien_weights(cfg, seed=5, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(1)
    seq, dense, cat = _inputs(B, rs, cfg['category_hash_size'])

    def run(kernels):
        net = DeviceDien(dict(cfg, scorer_kernels=kernels), w, max_rows=B, max_slots=B + 1)
        net.encode(0, torch.from_numpy(np.ascontiguousarray(seq[:, 0])).cuda(), 0)
        net.encode(1, torch.zeros((1, 64), dtype=torch.int32).cuda(), B)
        sl = torch.full((2, B), B, dtype=torch.int32).cuda()
        sl[0] = torch.arange(B, dtype=torch.int32).cuda()
        obs, p = net.forward(B, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), sl.contiguous(), True, True)
        obs, p = obs.clone(), p.clone()
        net.close()
        return obs, p

    o_ref, p_ref = run('')
    o_two, p_two = run('no_dense_chain')
    assert torch.equal(o_ref, o_two) and torch.equal(p_ref, p_two)
    o_sep, p_sep = run(['no_head_fused'])
    assert (o_ref - o_sep).abs().max().item() < 2e-5 and (p_ref - p_sep).abs().max().item() < 2e-6


@pytest.mark.parametrize('variant', [dict(maxlen=33, category_feature_num=13, hidden_units=96, dense_feature_num=61, seq_num=3, class_num=3),
                                     dict(maxlen=64, category_feature_num=7, hidden_units=32, dense_feature_num=432, seq_num=1, class_num=2),
                                     dict(maxlen=16, category_feature_num=32, hidden_units=128, dense_feature_num=40, seq_num=2, class_num=2)])
def test_dien_other_configurations(variant):
    """Non-default DIEN shapes (sequence length, category count, tower width, number of sequence inputs, classes)."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from oracle.dien import OracleDien
    cfg = dict(CFG, **variant)
    L, Cn, Dn, S = cfg['maxlen'], cfg['category_feature_num'], cfg['dense_feature_num'], cfg['seq_num']
    w = init_dien_weights(cfg, seed=5, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(7)
    R = 45
    seq = rs.randint(0, 284, size=(R, S, L)).astype(np.int32)
    dense = np.abs(rs.randn(R, Dn) * 3).astype(np.float32)
    cat = rs.randint(0, cfg['category_hash_size'], size=(R, Cn)).astype(np.int32)
    net = DeviceDien(cfg, w, max_rows=R, max_slots=R)
    for s in range(S):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    slots = torch.arange(R, dtype=torch.int32).repeat(S, 1).contiguous().cuda()
    obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, True, True)
    orc = OracleDien(w, cfg, np.float64)
    assert np.abs(obs.cpu().numpy() - orc.obs(seq, dense, cat)).max() < 5e-5
    assert np.abs(prob.cpu().numpy() - orc.reward_probs(seq, dense, cat)[:, 1]).max() < 5e-6
    net.close()


def test_scorer_mode_selection(monkeypatch, scorer_precision):
    """scorer_precision / scorer_kernels / RL4RS_SCORER (a default the PYTHON layer reads; the library never looks at the
    environment) and the rule of rl4rs_dien_create: any finite checkpoint runs in fp16x2 (power-of-two prescale per matrix)."""
    if scorer_precision != 'fp32':
        pytest.skip('mode-independent')
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from rl4rs_amd._lib import Rl4rsHipError
    w = init_dien_weights(CFG, seed=1)
    base = dict(CFG)
    base.pop('scorer_precision')
    monkeypatch.delenv('RL4RS_SCORER', raising=False)
    monkeypatch.delenv('RL4RS_DIEN_OPTS', raising=False)
    net = DeviceDien(base, w, max_rows=8, max_slots=4)
    assert net.scorer_mode == 'fp16x2' and net.augru_kernel == 'k_augru_x'        # auto default: second-generation recurrence
    net.close()
    net = DeviceDien(dict(base, scorer_kernels='augru_h16'), w, max_rows=8, max_slots=4)    # the first generation stays selectable
    assert net.scorer_mode == 'fp16x2' and net.augru_kernel == 'k_augru_h16'
    net.close()
    monkeypatch.setenv('RL4RS_SCORER', 'fp32')
    net = DeviceDien(base, w, max_rows=8, max_slots=4)
    assert net.scorer_mode == 'fp32' and net.augru_kernel == 'k_recur<256,augru>'
    net.close()
    net = DeviceDien(dict(base, scorer_precision='fp16x2'), w, max_rows=8, max_slots=4)   # explicit beats env
    assert net.scorer_mode == 'fp16x2'
    net.close()
    monkeypatch.setenv('RL4RS_SCORER', 'bf16')
    with pytest.raises(ValueError, match='scorer_precision'):
        DeviceDien(base, w, max_rows=8, max_slots=4)
    monkeypatch.delenv('RL4RS_SCORER')
    with pytest.raises(ValueError, match='scorer_precision'):
        DeviceDien(dict(base, scorer_precision='tf32'), w, max_rows=8, max_slots=4)
    # weights outside the fp16 range: fp16x2 is KEPT (each matrix is stored times its own power of two)
    big = dict(w)
    big['augru0_cand_w'] = w['augru0_cand_w'].copy()
    big['augru0_cand_w'][5, 7] = 7.0e4
    big['augru1_gate_w'] = w['augru1_gate_w'].copy()
    big['augru1_gate_w'][200, 300] = -3.0e5
    net = DeviceDien(base, big, max_rows=8, max_slots=4)
    assert net.scorer_mode == 'fp16x2'
    net.close()
    net = DeviceDien(dict(base, scorer_precision='fp16x2'), big, max_rows=8, max_slots=4)
    assert net.scorer_mode == 'fp16x2'
    net.close()
    # only a non-finite weight cannot be carried
    bad = dict(w)
    bad['augru0_cand_w'] = w['augru0_cand_w'].copy()
    bad['augru0_cand_w'][5, 7] = np.inf
    net = DeviceDien(base, bad, max_rows=8, max_slots=4)
    assert net.scorer_mode == 'fp32'
    net.close()
    with pytest.raises(Rl4rsHipError, match='finite'):
        DeviceDien(dict(base, scorer_precision='fp16x2'), bad, max_rows=8, max_slots=4)


@pytest.mark.parametrize('variant', ['outliers', 'tiny', 'gru_outlier'])
def test_split_form_is_total_over_weight_scales(variant, scorer_precision):
    """the fp16x2 form must not depend on the scale of a checkpoint.  'outliers': recurrent AUGRU weights of
    7e4 / -3e5 and x-side / head weights of 1e5 / 9e4 (all beyond fp16; weights that would push an ACTIVATION beyond fp16
    are a different matter: that row is NaN-poisoned and reported) - the handle stays in fp16x2 and every
    intermediate and output meets the SAME bars against the fp64 oracle; 'tiny': every AUGRU weight times 2^-12 (unscaled,
    their fp16 hi parts would be subnormal: 3 % relative error per weight); 'gru_outlier': a first-GRU weight beyond fp16
    (that piece alone falls back to its exact-fp32 kernel, the rest stays split)."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only')
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien, DIEN_ALL_FEATURE
    from oracle.dien import OracleDien
    R = 96
    w = init_dien_weights(CFG, seed=21, emb_scale=0.5, bias_noise=0.2)
    w = dict((k, np.array(v, copy=True)) for k, v in w.items())
    if variant == 'outliers':
        w['augru0_cand_w'][128 + 5, 7] = 7.0e4           # h-side rows start at E = 128: a saturated candidate column
        w['augru1_gate_w'][128 + 200, 300] = -3.0e5      # a saturated update-gate column
        w['augru0_gate_w'][3, 40] = 1.0e5                # x-side (projection GEMM)
        w['obs_w'][600, 9] = 9.0e4                       # head GEMM, on a dense-tower feature (bounded by ELU inputs)
    elif variant == 'tiny':
        for s in range(2):
            w['augru%d_gate_w' % s] *= np.float32(2.0 ** -12)
            w['augru%d_cand_w' % s] *= np.float32(2.0 ** -12)
    else:
        w['gru0_cand_w'][128 + 9, 11] = 8.0e4
    rs = np.random.RandomState(5)
    seq, dense, cat = _inputs(R, rs, CFG['category_hash_size'])
    net = DeviceDien(CFG, w, max_rows=R, max_slots=R)
    assert net.scorer_mode == 'fp16x2'
    for s in range(2):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    slots = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
    obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, want_obs=True, want_prob=True)
    orc = OracleDien(w, CFG, np.float64)
    allf, parts = orc.features(seq, dense, cat, return_parts=True)
    af = net.snapshot(DIEN_ALL_FEATURE, R).cpu().numpy()[:R]
    assert np.abs(af[:, :256] - parts['h2_0']).max() < 5e-6
    assert np.abs(af[:, 256:512] - parts['h2_1']).max() < 5e-6
    obs_ref = orc.obs(seq, dense, cat)
    tol = 5e-5 * max(1.0, np.abs(obs_ref).max() / 10.0)      # the 9e4 head weight makes some activations O(1e4)
    assert np.abs(obs.cpu().numpy() - obs_ref).max() < tol
    assert np.abs(prob.cpu().numpy() - orc.reward_probs(seq, dense, cat)[:, 1]).max() < 5e-6
    net.check_status()
    net.close()


def test_fp16_range_status(scorer_precision):
    """RL4RS_DIEN_STATUS_FP16_RANGE: attention scores far below 0 make (1 - a_t) u exceed 1, the AUGRU state then grows
    geometrically; the fp16x2 kernel must report it (the fp32 kernel has nothing to report), and a sane model must not."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from rl4rs_amd._lib import Rl4rsHipError
    R = 40
    rs = np.random.RandomState(2)
    seq, dense, cat = _inputs(R, rs, CFG['category_hash_size'])

    def run(w):
        net = DeviceDien(CFG, w, max_rows=R, max_slots=R)
        for s in range(2):
            net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
        slots = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
        net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, want_obs=True)
        return net

    net = run(init_dien_weights(CFG, seed=3, emb_scale=0.5))
    net.check_status()                                  # sane model: silent in both modes
    net.close()
    bad = init_dien_weights(CFG, seed=3, emb_scale=0.5)
    for i in range(2):
        bad['att%d_b3' % i] = np.array([-40.0], dtype=np.float32)      # a_t ~ -40: u <- 41 u
    net = run(bad)
    if scorer_precision == 'fp16x2':
        with pytest.raises(Rl4rsHipError, match='fp16 range'):
            net.check_status()
        net.check_status()                              # the flag is cleared by the read
    else:
        net.check_status()
    net.close()


def test_fp16_range_poisons_the_rows_on_the_device(scorer_precision):
    """in tensor mode nobody polls the status bit, so an out-of-range recurrence must not hand back
    plausible numbers.  k_augru_x writes NaN into every output column of a row whose state left the fp16 range: its
    observation and click probability are NaN ON THE DEVICE (no host synchronisation), the rows of a sane model are not."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only')
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    R = 40
    rs = np.random.RandomState(2)
    seq, dense, cat = _inputs(R, rs, CFG['category_hash_size'])
    w = init_dien_weights(CFG, seed=3, emb_scale=0.5)
    w['att0_b3'] = np.array([-40.0], dtype=np.float32)          # sequence input 0 explodes for every row
    net = DeviceDien(CFG, w, max_rows=R, max_slots=R)
    for s in range(2):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    slots = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
    obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, want_obs=True, want_prob=True)
    assert torch.isnan(obs).all(dim=1).all() and torch.isnan(prob).all()
    net.close()
    ok = init_dien_weights(CFG, seed=3, emb_scale=0.5)
    net = DeviceDien(CFG, ok, max_rows=R, max_slots=R)
    for s in range(2):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    obs, prob = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, want_obs=True, want_prob=True)
    assert torch.isfinite(obs).all() and torch.isfinite(prob).all()
    net.check_status()
    net.close()


def test_first_generation_recurrence_still_matches(monkeypatch, scorer_precision):
    """scorer_kernels='augru_h16' (k_augru_h16, round 1) stays a selectable fp16x2 recurrence and stays parity-green."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only')
    monkeypatch.setitem(CFG, 'scorer_kernels', 'augru_h16')
    test_dien_rowwise_matches_oracle(64)


@pytest.mark.parametrize('kernels', ['din_v1', 'no_gru16', 'no_gemm16,no_cat16', 'no_head_tables', 'cat_v1', 'cat_v1,no_cat16',
                                     'no_cat16', 'cat_v1,no_head_tables'])
def test_other_kernel_paths_match_the_oracle(monkeypatch, scorer_precision, kernels):
    """Every selectable kernel path of rl4rs_dien_cfg.kernel_opts against the same fp64 oracle and the same bars."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only')
    monkeypatch.setitem(CFG, 'scorer_kernels', kernels)
    test_dien_rowwise_matches_oracle(64)


@pytest.mark.parametrize('extra', ['', 'no_cat16', 'no_head_tables'])
def test_category_kernels_are_bit_identical(scorer_precision, extra):
    """k_cat_attn2 (half-K LDS image, pooled row from the gathered registers, staged table sums) against k_cat_attn
    (scorer_kernels='cat_v1'): same products in the same order - the whole forward must be bit-identical, in the split and the
    exact-fp32 form of the Gram matrix and with the flattened embeddings written out (no head tables)."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    cfg = dict(CFG, scorer_precision=scorer_precision)
    B = 101
    w = init_dien_weights(cfg, seed=6, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(2)
    seq, dense, cat = _inputs(B, rs, cfg['category_hash_size'])

    def run(kernels):
        net = DeviceDien(dict(cfg, scorer_kernels=kernels), w, max_rows=B, max_slots=B)
        for s_ in range(2):
            net.encode(s_, torch.from_numpy(np.ascontiguousarray(seq[:, s_])).cuda(), 0)
        sl = torch.arange(B, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
        obs, p = net.forward(B, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), sl, True, True)
        obs, p = obs.clone(), p.clone()
        net.close()
        return obs, p

    o2, p2 = run(extra)
    o1, p1 = run(','.join(x for x in ('cat_v1', extra) if x))
    assert torch.equal(o1, o2) and torch.equal(p1, p2)


@pytest.mark.parametrize('group', [8, 9])
@pytest.mark.parametrize('extra', ['', 'no_head_tables', 'no_cat16'])
def test_group_category_kernel_is_bit_identical(scorer_precision, group, extra):
    """k_cat_attn2g (one workgroup per group of rows that share a cache slot: the category rows the group shares gathered once,
    sums in the per-row kernel's order) against k_cat_attn2 (scorer_kernels='no_cat_group') on a reward-shaped launch: groups of
    complete states (rl4rs/env/slate.py:117-131: every id but the last shared), groups where a few rows differ in a shared id
    (extra passes of the kernel's loop), one group that shares nothing, ids outside the table (clamped alike) - the whole forward
    must be bit-identical."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    cfg = dict(CFG, scorer_precision=scorer_precision)
    B = 37
    R = B * group
    w = init_dien_weights(cfg, seed=8, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(5)
    seq, dense, _ = _inputs(B, rs, cfg['category_hash_size'])
    dense = np.abs(rs.randn(R, 432) * 3).astype(np.float32)
    _, _, cat_env = _inputs(B, rs, cfg['category_hash_size'])
    cat = np.repeat(cat_env, group, axis=0).reshape(B, group, 21)
    cat[:, :, 20] = rs.randint(0, 284, size=(B, group))             # the row's own item
    cat[3, 2, 5] += 1                                                # one row of a group differs in a shared id
    cat[4, 1::2, 12] = 7                                             # every second row of a group: two sets of rows
    cat[5] = rs.randint(0, 284, size=(group, 21))                    # a group that shares nothing
    cat[6, :, 0] = cfg['category_hash_size'] + 5                     # outside the table: clamped by both kernels
    cat[7, 3, 19] = -2
    cat = np.ascontiguousarray(cat.reshape(R, 21)).astype(np.int32)

    def run(kernels):
        net = DeviceDien(dict(cfg, scorer_kernels=kernels), w, max_rows=R, max_slots=B)
        for s_ in range(2):
            net.encode(s_, torch.from_numpy(np.ascontiguousarray(seq[:, s_])).cuda(), 0)
        sl = torch.arange(B, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
        obs, p = net.forward(R, group, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), sl, True, True)
        obs, p = obs.clone(), p.clone()
        net.close()
        return obs, p

    og, pg = run(extra)
    o1, p1 = run(','.join(x for x in ('no_cat_group', extra) if x))
    assert torch.isfinite(og).all() and torch.isfinite(pg).all()
    assert torch.equal(o1, og) and torch.equal(p1, pg)


def test_kernel_options_are_validated():
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from rl4rs_amd._lib import Rl4rsHipError
    w = init_dien_weights(CFG, seed=1)
    with pytest.raises(ValueError, match='scorer_kernels'):
        DeviceDien(dict(CFG, scorer_kernels='augru_rows48'), w, max_rows=8, max_slots=4)
    with pytest.raises(Rl4rsHipError, match='32-row and the 64-row'):
        DeviceDien(dict(CFG, scorer_kernels='augru_rows32,augru_rows64'), w, max_rows=8, max_slots=4)
    net = DeviceDien(CFG, w, max_rows=8, max_slots=4)
    with pytest.raises(Rl4rsHipError, match='rows must be'):
        net.set_augru_rows(48)
    net.close()


@pytest.mark.parametrize('B', [64, 136])
def test_augru_64_row_form_every_row_against_the_oracle(B, scorer_precision):
    """The 64-row form of k_augru_x (two row tiles per weight fragment - what the reward forward of the headline batch runs,
    41 % of the AUGRU time) PINNED with scorer_kernels='augru_rows64' at launch sizes where the automatic rule would pick the
    32-row form: R = 8 B rows in groups of 8 per cache slot (512 and 1088 rows), EVERY row's AUGRU final states (<= 5e-6),
    observation (<= 5e-5) and click probability (<= 5e-6) against the fp64 oracle; then the same cache scored by the 32-row
    form on the same handle (rl4rs_dien_set_augru_rows) and under a row-order permutation: bit-identical."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only')
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien, DIEN_ALL_FEATURE
    from oracle.dien import OracleDien
    G = 8
    R = B * G
    w = init_dien_weights(CFG, seed=13, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(B)
    seq_env, _, _ = _inputs(B, rs, CFG['category_hash_size'])
    seq_env[:, 1, :] = 0
    _, dense, cat = _inputs(R, rs, CFG['category_hash_size'])
    net = DeviceDien(dict(CFG, scorer_kernels='augru_rows64'), w, max_rows=R, max_slots=B + 1)
    net.encode(0, torch.from_numpy(np.ascontiguousarray(seq_env[:, 0])).cuda(), 0)
    net.encode(1, torch.zeros((1, 64), dtype=torch.int32).cuda(), B)
    sl = torch.full((2, B), B, dtype=torch.int32).cuda()
    sl[0] = torch.arange(B, dtype=torch.int32).cuda()
    sl = sl.contiguous()
    d, c = torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda()
    obs64, p64 = net.forward(R, G, d, c, sl, True, True)
    af64 = net.snapshot(DIEN_ALL_FEATURE, R)[:R, :512].clone()
    seq_rows = np.repeat(seq_env, G, axis=0)
    orc = OracleDien(w, CFG, np.float64)
    _, parts = orc.features(seq_rows, dense, cat, return_parts=True)
    a = af64.cpu().numpy()
    assert np.abs(a[:, :256] - parts['h2_0']).max() < 5e-6
    assert np.abs(a[:, 256:512] - parts['h2_1']).max() < 5e-6
    assert np.abs(obs64.cpu().numpy() - orc.obs(seq_rows, dense, cat)).max() < 5e-5
    assert np.abs(p64.cpu().numpy() - orc.reward_probs(seq_rows, dense, cat)[:, 1]).max() < 5e-6
    # the 32-row form over the same cache: same MFMA sequence per row
    net.set_augru_rows(32)
    obs32, p32 = net.forward(R, G, d, c, sl, True, True)
    af32 = net.snapshot(DIEN_ALL_FEATURE, R)[:R, :512].clone()
    assert torch.equal(af32, af64) and torch.equal(obs32, obs64) and torch.equal(p32, p64)
    # a processing order (locality hint) must not change any row, in either form
    order = torch.from_numpy(np.random.RandomState(1).permutation(B).astype(np.int32)).cuda()
    net.set_row_order(order)
    for rows in (64, 32):
        net.set_augru_rows(rows)
        o, p = net.forward(R, G, d, c, sl, True, True)
        assert torch.equal(o, obs64) and torch.equal(p, p64), rows
    net.check_status()
    net.close()


def test_fp16x2_recurrence_with_fp32_attention(monkeypatch, scorer_precision):
    """fp16x2 AUGRU combined with the exact-fp32 DIN layer 1 (what the library falls back to when the sequence embedding
    table or the q*k rows of att_w1 leave the fp16 range; forced here with scorer_kernels='no_din16')."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only')
    monkeypatch.setitem(CFG, 'scorer_kernels', 'no_din16')
    test_dien_rowwise_matches_oracle(64)


@pytest.mark.parametrize('L', [8, 16, 33])
def test_short_histories_stay_in_bounds(L, scorer_precision):
    """maxlen < 32: the DIN score kernel used to read first-GRU cache rows past a slot's L steps (lanes 16..31 of its
    32-step tile); harmless inside a larger allocation, a memory fault when the cache ends on a page boundary
    (maxlen 16 x 256 slots x 128 floats = exactly 2 MB).  All three recurrences, whole forward vs the fp64 oracle."""
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from oracle.dien import OracleDien
    B = 256
    cfg = dict(CFG, maxlen=L, batch_size=B)
    w = init_dien_weights(cfg, seed=7, emb_scale=0.5, bias_noise=0.1)
    rs = np.random.RandomState(L)
    seq = rs.randint(0, 284, size=(B, 2, L)).astype(np.int32)
    dense = np.abs(rs.randn(B, cfg['dense_feature_num'])).astype(np.float32)
    cat = rs.randint(0, 284, size=(B, cfg['category_feature_num'])).astype(np.int32)
    slots = torch.arange(B, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
    net = DeviceDien(cfg, w, max_rows=B, max_slots=B)
    for s in range(2):
        net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
    obs, _ = net.forward(B, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), slots, want_obs=True, want_prob=False)
    ref = OracleDien(w, cfg, np.float64).obs(seq, dense, cat)
    assert np.abs(obs.cpu().numpy() - ref).max() < 5e-5
    net.check_status()
    net.close()


def test_trained_like_model_stays_within_the_margins(scorer_precision):
    """Both recurrences on TRAINED-like weights: 150 Adam steps of the device DIEN trainer on synthetic logs,
    then the forward against the fp64 oracle on those weights - raw attention scores of both signs, same parity bars as the
    synthetic-weight tests (obs 5e-5 abs, click probability 5e-6 abs), and no fp16-range status."""
    import importlib.util
    import os
    if scorer_precision != 'fp16x2':
        pytest.skip('runs both modes itself')
    spec = importlib.util.spec_from_file_location('error_margins', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'error_margins.py'))
    # the tool prints its synthetic-weight table at import: run only the function
    src = open(spec.origin).read()
    ns = {}
    head, tail = src.split("\n\ndef trained_like", 1)
    exec(compile(head.split("for name, kw in")[0] + "\n\ndef trained_like" + tail.split("\n\nif __name__")[0], spec.origin, 'exec'), ns)
    out = ns['trained_like'](steps=150, R=256, verbose=False)
    assert out['loss_last'] < out['loss_first']
    wide = out['variants']['trained, attention head centred and widened to std 0.1']
    assert wide['score_min'] < -0.15 and wide['score_max'] > 0.15                     # (1 - a_t) on both sides of 1
    for name, v in out['variants'].items():
        for mode in ('fp32', 'fp16x2'):
            # rows the fp16x2 recurrence could not carry are NaN + status bit (never a plausible wrong number); every other row
            # meets the parity bars in both modes
            assert v[mode]['flagged'] == (v[mode]['poisoned_rows'] > 0)
            assert mode == 'fp16x2' or v[mode]['poisoned_rows'] == 0
            assert v[mode]['obs_abs'] < 5e-5 and v[mode]['prob_abs'] < 5e-6, (name, mode, out)
    assert out['variants']['trained']['fp16x2']['poisoned_rows'] == 0


def test_first_gru_pad_table_is_bit_identical(scorer_precision):
    """k_gru_h16 takes the steps on LEADING zero ids (pad_sequences pads in front, rl4rs/utils/datautil.py:44) from the handle's
    table of pad states when all 32 rows of a workgroup share them (RecurArgs::pad), instead of computing them per row
    (scorer_kernels='no_gru_pad').  Workgroups of every kind - a common prefix of 55 / 46 / 37 zeros as SeqSlate's second input
    has at its page ends, mixed prefix lengths, all-zero rows only, no prefix at all, zeros in the MIDDLE of a row, a ragged last
    workgroup - must give the same h1 cache and the same forward, bit for bit."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only: the exact-fp32 first GRU has no pad table')
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien, DIEN_H1
    cfg = dict(CFG, scorer_precision=scorer_precision)
    w = init_dien_weights(cfg, seed=8, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(11)
    blocks = []
    for lead in (55, 46, 37):                                # one workgroup each: the same prefix on every row
        b = rs.randint(1, 284, size=(32, 64)); b[:, :lead] = 0; blocks.append(b)
    b = rs.randint(1, 284, size=(32, 64))                    # mixed prefixes, the shortest is 7
    for r in range(32):
        b[r, :7 + (r * 5) % 50] = 0
    blocks.append(b)
    blocks.append(np.zeros((32, 64), dtype=np.int64))        # nothing but padding
    blocks.append(rs.randint(1, 284, size=(32, 64)))         # no padding
    b = rs.randint(1, 284, size=(32, 64)); b[:, :12] = 0; b[:, 30:40] = 0; b[3, 12] = 0; blocks.append(b)    # zeros further in are ordinary steps
    b = rs.randint(1, 284, size=(13, 64)); b[:, :60] = 0; blocks.append(b)                                   # ragged tail
    seq1 = np.concatenate(blocks).astype(np.int32)
    R = seq1.shape[0]
    seq0 = rs.randint(0, 284, size=(R, 64)).astype(np.int32)
    seq0[: R // 2, :33] = 0
    dense = np.abs(rs.randn(R, 432) * 3).astype(np.float32)
    cat = rs.randint(0, cfg['category_hash_size'], size=(R, 21)).astype(np.int32)
    cat[:, 10:] = rs.randint(0, 284, size=(R, 11))

    def run(kernels):
        net = DeviceDien(dict(cfg, scorer_kernels=kernels), w, max_rows=R, max_slots=R + 40)
        net.encode(0, torch.from_numpy(seq0).cuda(), 0)
        net.encode(1, torch.from_numpy(seq1).cuda(), 0)
        net.encode(1, torch.from_numpy(np.ascontiguousarray(seq1[:40])).cuda(), R)        # a second call at a slot offset
        h1 = net.snapshot(DIEN_H1, R)[:R].clone()            # (sequence input 0's cache; the slots behind R are never written)
        sl = torch.arange(R, dtype=torch.int32).repeat(2, 1).contiguous().cuda()
        obs, p = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), sl, True, True)
        obs, p = obs.clone(), p.clone()
        net.close()
        return h1, obs, p

    h1a, oa, pa = run('')
    h1b, ob, pb = run('no_gru_pad')
    assert torch.isfinite(oa).all() and torch.isfinite(pa).all()
    assert torch.equal(h1a, h1b) and torch.equal(oa, ob) and torch.equal(pa, pb)


@pytest.mark.parametrize('L', [33, 16])
def test_pad_table_at_other_sequence_lengths(scorer_precision, L):
    """The pad-state table / pad slot (test above) at maxlen 33 and 16 (one ragged 32-step tile, half a tile) with three sequence
    inputs: front-padded rows against the fp64 oracle and, bit for bit, against scorer_kernels='no_gru_pad'."""
    if scorer_precision != 'fp16x2':
        pytest.skip('fp16x2 only: the exact-fp32 first GRU has no pad table')
    import torch
    from rl4rs_amd.nets.dien import init_dien_weights
    from rl4rs_amd.device import DeviceDien
    from oracle.dien import OracleDien
    cfg = dict(CFG, scorer_precision=scorer_precision, maxlen=L, seq_num=3)
    w = init_dien_weights(cfg, seed=9, emb_scale=0.5, bias_noise=0.2)
    rs = np.random.RandomState(L)
    R, S = 70, 3
    seq = rs.randint(1, 284, size=(R, S, L)).astype(np.int32)
    for r in range(R):
        seq[r, 0, :(r * 3) % (L + 1)] = 0                     # every prefix length incl. 0 and L
        seq[r, 1, :L - 2] = 0                                 # the same long prefix on every row
    seq[:, 2, :] = 0                                          # a constant all-zero input
    dense = np.abs(rs.randn(R, 432) * 3).astype(np.float32)
    cat = rs.randint(0, cfg['category_hash_size'], size=(R, 21)).astype(np.int32)

    def run(kernels):
        net = DeviceDien(dict(cfg, scorer_kernels=kernels), w, max_rows=R, max_slots=R)
        for s in range(S):
            net.encode(s, torch.from_numpy(np.ascontiguousarray(seq[:, s])).cuda(), 0)
        sl = torch.arange(R, dtype=torch.int32).repeat(S, 1).contiguous().cuda()
        obs, p = net.forward(R, 1, torch.from_numpy(dense).cuda(), torch.from_numpy(cat).cuda(), sl, True, True)
        obs, p = obs.clone(), p.clone()
        net.close()
        return obs, p

    oa, pa = run('')
    ob, pb = run('no_gru_pad')
    assert torch.equal(oa, ob) and torch.equal(pa, pb)
    orc = OracleDien(w, cfg, np.float64)
    assert np.abs(oa.cpu().numpy() - orc.obs(seq, dense, cat)).max() < 5e-5
    assert np.abs(pa.cpu().numpy() - orc.reward_probs(seq, dense, cat)[:, 1]).max() < 5e-6
