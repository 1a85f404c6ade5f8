"""GPU parity of the HIP policy net: forward / masking / sampling / A2C + PPO gradients against the fp64
restatement (oracle/policy.py, torch autograd).  Tolerances: logits 1e-5 abs, gradients 2e-4 relative to the
gradient's max-norm (fp32 kernels vs fp64 autograd over thousands of samples)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(N, rs, A=284):
    obs = rs.randn(N, 256).astype(np.float32)
    mask = (rs.rand(N, A) < 0.4).astype(np.int64)
    mask[np.arange(N), rs.randint(0, A, size=N)] = 1        # at least one allowed action per row
    mask[0] = 1
    W = (A + 31) // 32
    bits = np.zeros((N, W), dtype=np.uint32)
    for k in range(A):
        bits[:, k >> 5] |= (mask[:, k].astype(np.uint32) << np.uint32(k & 31))
    return obs, mask, bits.view(np.int32)


def test_forward_masking_and_sampling():
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    from oracle import policy as OP
    rs = np.random.RandomState(0)
    N = 1003                                                 # not a multiple of the 8-row tiles of k_policy_tile
    obs, mask, bits = _data(N, rs)
    flat = init_policy_params(seed=3) + (rs.randn(34973) * 0.05).astype(np.float32)
    pol = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
    assert pol.n_params == 34973
    o, b = torch.from_numpy(obs).cuda(), torch.from_numpy(bits).cuda()
    a, lp, v, ent, lg = pol.act(o, b, seed=5, step=7, want_logits=True)
    logits, value = OP.forward(flat, obs, mask)
    a_np = a.cpu().numpy()
    allowed = mask[np.arange(N), a_np]
    assert allowed.all()                                      # never samples a masked action
    lg_np = lg.cpu().numpy()
    ok = mask > 0
    assert np.abs(lg_np[ok] - logits[ok]).max() < 1e-5
    assert (lg_np[~ok] < -1e37).all()
    lsm = OP.log_softmax(logits)
    assert np.abs(lp.cpu().numpy() - lsm[np.arange(N), a_np]).max() < 2e-5
    assert np.abs(v.cpu().numpy() - value).max() < 1e-5
    p = np.exp(lsm)
    ent_ref = -np.where(p > 0, p * lsm, 0).sum(1)
    assert np.abs(ent.cpu().numpy() - ent_ref).max() < 2e-5
    # deterministic for a (seed, step) pair, different across steps
    a2 = pol.act(o, b, seed=5, step=7)[0]
    a3 = pol.act(o, b, seed=5, step=8)[0]
    assert torch.equal(a, a2) and not torch.equal(a, a3)
    # evaluate() reproduces the sampled log-probs; no mask = all allowed
    lp2, v2, _, _ = pol.evaluate(o, a, b)
    assert torch.equal(lp2, lp) and torch.equal(v2, v)
    lg_nomask = pol.act(o, None, seed=1, step=1, want_logits=True)[4].cpu().numpy()
    assert np.abs(lg_nomask - OP.forward(flat, obs, None)[0]).max() < 1e-5
    # sampling frequencies follow softmax (one row replicated)
    rep = np.repeat(obs[:1], 20000, axis=0)
    brep = np.repeat(bits[:1], 20000, axis=0)
    pol2 = DevicePolicy(256, 64, 284, max_rows=20000, params=flat * 30)
    aa = pol2.act(torch.from_numpy(rep).cuda(), torch.from_numpy(brep).cuda(), seed=11, step=0)[0].cpu().numpy()
    pr = np.exp(OP.log_softmax(OP.forward(flat * 30, obs[:1], mask[:1])[0]))[0]
    freq = np.bincount(aa, minlength=284) / 20000.0
    assert np.abs(freq - pr).max() < 0.02


@pytest.mark.parametrize('algo', [0, 1])
def test_loss_gradients_match_autograd(algo):
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    from oracle import policy as OP
    rs = np.random.RandomState(algo + 1)
    N = 1500
    obs, mask, bits = _data(N, rs)
    flat = init_policy_params(seed=1) + (rs.randn(34973) * 0.05).astype(np.float32)
    old = flat + (rs.randn(34973) * 0.01).astype(np.float32)
    pol = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
    o, b = torch.from_numpy(obs).cuda(), torch.from_numpy(bits).cuda()
    old_logits, old_value = OP.forward(old, obs, mask)
    old_lsm = OP.log_softmax(old_logits)
    actions = np.array([rs.choice(np.nonzero(mask[i])[0]) for i in range(N)])
    old_logp = old_lsm[np.arange(N), actions]
    adv = rs.randn(N) * 3
    ret = rs.randn(N) * 50 + 100
    kw = dict(vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=30.0, kl_coeff=0.2)
    g, stats = pol.loss_grad(algo, o, torch.from_numpy(actions).cuda(), torch.from_numpy(adv).cuda(),
                             torch.from_numpy(ret).cuda(), mask_bits=b,
                             old_logp=torch.from_numpy(old_logp).cuda(), old_value=torch.from_numpy(old_value).cuda(),
                             old_logits=torch.from_numpy(np.maximum(old_logits, -3.4e38).astype(np.float32)).cuda(), **kw)
    g_ref, s_ref = OP.loss_and_grad(algo, flat, obs, mask, actions, adv, ret, old_logp, old_value, old_logits, **kw)
    g_np = g.cpu().numpy()
    assert np.abs(g_np - g_ref).max() < 2e-4 * np.abs(g_ref).max(), (np.abs(g_np - g_ref).max(), np.abs(g_ref).max())
    assert np.allclose(stats.cpu().numpy(), s_ref, rtol=2e-4, atol=1e-3)
    # bit-reproducible
    g2, _ = pol.loss_grad(algo, o, torch.from_numpy(actions).cuda(), torch.from_numpy(adv).cuda(),
                          torch.from_numpy(ret).cuda(), mask_bits=b, old_logp=torch.from_numpy(old_logp).cuda(),
                          old_value=torch.from_numpy(old_value).cuda(),
                          old_logits=torch.from_numpy(np.maximum(old_logits, -3.4e38).astype(np.float32)).cuda(), **kw)
    assert torch.equal(g, g2)
    # Adam step (tf AdamOptimizer form) + global-norm clipping
    before = pol.params().cpu().numpy().astype(np.float64)
    pol.adam_step(g, lr=1e-3, grad_clip=10.0)
    after = pol.params().cpu().numpy().astype(np.float64)
    gn = np.sqrt((g_np.astype(np.float64) ** 2).sum())
    gc = g_np * min(1.0, 10.0 / gn)
    m = 0.1 * gc
    vv = 0.001 * gc * gc
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    expect = before - lr_t * m / (np.sqrt(vv) + 1e-8)
    assert np.abs(after - expect).max() < 1e-6


def test_ppo_epoch_equals_minibatch_sequence():
    """rl4rs_policy_ppo_epoch (one persistent kernel for the whole pass, k_ppo_pass) == loss_grad + adam_step per minibatch
    (trailing rows dropped).  The fused pass sums its MFMA tiles in another order than the per-minibatch kernels, so the
    comparison is numerical: Adam turns a gradient into ~lr * sign-like steps, hence entries whose gradient is at rounding
    level may differ by up to lr per step; everything else must agree to 2e-5."""
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    from oracle import policy as OP
    rs = np.random.RandomState(9)
    N, MB = 1100, 256
    obs, mask, bits = _data(N, rs)
    flat = init_policy_params(seed=2) + (rs.randn(34973) * 0.05).astype(np.float32)
    old_logits, old_value = OP.forward(flat, obs, mask)
    old_lsm = OP.log_softmax(old_logits)
    actions = np.array([rs.choice(np.nonzero(mask[i])[0]) for i in range(N)])
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    o, b, a = t(obs), torch.from_numpy(bits).cuda(), t(actions, torch.int32)
    adv, ret = t(rs.randn(N) * 2), t(rs.randn(N) * 50 + 100)
    olp, ov = t(old_lsm[np.arange(N), actions]), t(old_value)
    ol = t(np.maximum(old_logits, -3.4e38).astype(np.float32))
    kw = dict(vf_coeff=0.5, ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=0.2)
    p1 = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
    p2 = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
    stats1 = None
    for lo in range(0, N - MB + 1, MB):
        hi = lo + MB
        g, stats1 = p1.loss_grad(1, o[lo:hi], a[lo:hi], adv[lo:hi], ret[lo:hi], mask_bits=b[lo:hi], old_logp=olp[lo:hi],
                                 old_value=ov[lo:hi], old_logits=ol[lo:hi], **kw)
        p1.adam_step(g, lr=1e-3)
    stats2 = p2.ppo_epoch(o, a, adv, ret, b, olp, ov, ol, minibatch=MB, lr=1e-3, **kw)
    w1, w2 = p1.params().cpu().numpy().astype(np.float64), p2.params().cpu().numpy().astype(np.float64)
    diff = np.abs(w1 - w2)
    steps = N // MB
    assert (diff < 2e-5).mean() > 0.999, (diff < 2e-5).mean()
    assert diff.max() <= 2 * steps * 1e-3, diff.max()
    moved = np.abs(w1 - flat.astype(np.float64))
    assert np.median(diff[moved > 1e-4]) < 1e-6
    assert np.allclose(stats1.cpu().numpy(), stats2.cpu().numpy()[:4], rtol=2e-3, atol=1e-4), (stats1, stats2)     # [4:8] = pass-wide sums
    assert not torch.equal(p1.params().cpu(), torch.from_numpy(flat))


@pytest.mark.parametrize('N,with_mask', [(4096, True), (1003, True), (517, False), (5, True)])
def test_tile_kernels_of_the_default_shape_match_the_generic_ones(N, with_mask):
    """k_policy_tile_std<0 / 1 / 2> (the default shape on 4-row x 64-column MFMA tiles, policy_tile_std.hpp) against k_policy_tile (32x32x2
    tiles, RL4RS_POLICY_OPT_PPO_STD = 0): act with the same seed (masked logits to 2e-5, identical draws wherever the top two Gumbel
    scores are not within rounding of each other - counted, at most a handful), evaluate, and the A2C / PPO gradient; ragged last
    workgroups (N % 8 != 0), with and without a mask; and act / evaluate against the float64 restatement."""
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    from oracle import policy as OP
    rs = np.random.RandomState(N)
    obs, mask, bits = _data(N, rs)
    if not with_mask:
        mask = np.ones_like(mask)
    flat = init_policy_params(seed=6) + (rs.randn(34973) * 0.05).astype(np.float32)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    o, b = t(obs), (torch.from_numpy(bits).cuda() if with_mask else None)
    std = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
    gen = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
    gen.set_option('ppo_std', 0)
    a1, lp1, v1, e1, lg1 = std.act(o, mask_bits=b, seed=11, step=3, want_logits=True)
    a2, lp2, v2, e2, lg2 = gen.act(o, mask_bits=b, seed=11, step=3, want_logits=True)
    ref_logits, ref_value = OP.forward(flat, obs, mask)
    live = mask.astype(bool)
    assert np.abs(lg1.cpu().numpy()[live] - ref_logits[live]).max() < 5e-5 and np.abs(v1.cpu().numpy() - ref_value).max() < 5e-5
    assert np.abs(lg1.cpu().numpy()[live] - lg2.cpu().numpy()[live]).max() < 2e-5
    assert (lg1.cpu().numpy()[~live] < -1e38).all()
    assert mask[np.arange(N), a1.cpu().numpy()].all()                                   # never a masked action
    assert (a1 != a2).sum().item() <= max(2, N // 500), (a1 != a2).sum().item()
    same = (a1 == a2).cpu().numpy()
    assert np.allclose(lp1.cpu().numpy()[same], lp2.cpu().numpy()[same], atol=2e-5) and torch.allclose(v1, v2, atol=2e-5) and torch.allclose(e1, e2, atol=2e-5)
    l1, w1, n1, _ = std.evaluate(o, a2, mask_bits=b)
    l2, w2, n2, _ = gen.evaluate(o, a2, mask_bits=b)
    assert torch.allclose(l1, l2, atol=2e-5) and torch.allclose(w1, w2, atol=2e-5) and torch.allclose(n1, n2, atol=2e-5)
    lsm = OP.log_softmax(ref_logits)
    assert np.abs(l1.cpu().numpy() - lsm[np.arange(N), a2.cpu().numpy()]).max() < 5e-5
    adv, ret = t(rs.randn(N) * 2), t(rs.randn(N) * 10 + 5)
    for algo, kw in ((0, dict(vf_coeff=0.5, ent_coeff=0.01)),
                     (1, dict(old_logp=l2, old_value=w2, old_logits=torch.clamp(lg2, min=-3.4e38), vf_coeff=0.5, ent_coeff=0.01, kl_coeff=0.2))):
        g1, s1 = std.loss_grad(algo, o, a2, adv, ret, mask_bits=b, **kw)
        g2, s2 = gen.loss_grad(algo, o, a2, adv, ret, mask_bits=b, **kw)
        scale = max(1.0, g2.abs().max().item())
        assert (g1 - g2).abs().max().item() < 2e-5 * scale * max(1, N // 256), (algo, (g1 - g2).abs().max().item(), scale)
        assert torch.allclose(s1, s2, rtol=2e-4, atol=2e-3), (s1, s2)


@pytest.mark.parametrize('MB,with_mask,rows', [(256, True, None), (512, True, None), (256, False, None), (256, True, 8), (512, True, 4)])
def test_ppo_pass_compile_time_instantiation_matches_the_runtime_one(MB, with_mask, rows):
    """k_ppo_pass<true> (the default shape as compile-time constants: 4-row x 64-column MFMA tiles with K split over the waves, the
    wave's Adam operands resident in registers) against k_pass<false> (RL4RS_POLICY_OPT_PPO_STD = 0: 32x32x2 tiles, all runtime) on
    the same pass: other summation orders, so numerical like pass-vs-chain (Adam turns rounding-level gradient entries into steps of
    up to lr), statistics to 2e-3; and the data-parallel form (gradient of one minibatch + rl4rs_policy_adam_step) of the
    compile-time instantiation is bit-identical to its own fused pass.  rows: 4- or 8-row workgroups of the compile-time form pinned
    (automatic: 4 at MB = 256 - 64 workgroups -, 8 at MB = 512)."""
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    from oracle import policy as OP
    rs = np.random.RandomState(19)
    N = 4 * MB + 37
    obs, mask, bits = _data(N, rs)
    if not with_mask:
        mask = np.ones_like(mask)
    flat = init_policy_params(seed=2) + (rs.randn(34973) * 0.05).astype(np.float32)
    old_logits, old_value = OP.forward(flat, obs, mask)
    old_lsm = OP.log_softmax(old_logits)
    actions = np.array([rs.choice(np.nonzero(mask[i])[0]) for i in range(N)])
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    o, b, a = t(obs), (torch.from_numpy(bits).cuda() if with_mask else None), t(actions, torch.int32)
    adv, ret = t(rs.randn(N) * 2), t(rs.randn(N) * 50 + 100)
    olp, ov = t(old_lsm[np.arange(N), actions]), t(old_value)
    ol = t(np.maximum(old_logits, -3.4e38).astype(np.float32))
    kw = dict(vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=500.0, kl_coeff=0.2)
    p_std, p_gen, p_dp = (DevicePolicy(256, 64, 284, max_rows=N, params=flat) for _ in range(3))
    p_gen.set_option('ppo_std', 0)
    if rows is not None:
        p_std.set_option('ppo_rows', rows)
        p_dp.set_option('ppo_rows', rows)
    s_std = s_gen = None
    for _ in range(2):                              # two passes: the Adam step counter and the resident moments carry over
        s_std = p_std.ppo_epoch(o, a, adv, ret, b, olp, ov, ol, minibatch=MB, lr=1e-3, **kw)
        s_gen = p_gen.ppo_epoch(o, a, adv, ret, b, olp, ov, ol, minibatch=MB, lr=1e-3, **kw)
    p_std.check_status()
    p_gen.check_status()
    w1, w2 = p_std.params().cpu().numpy().astype(np.float64), p_gen.params().cpu().numpy().astype(np.float64)
    diff = np.abs(w1 - w2)
    steps = 2 * (N // MB)
    assert (diff < 2e-5).mean() > 0.999, (diff < 2e-5).mean()
    assert diff.max() <= 2 * steps * 1e-3, diff.max()
    moved = np.abs(w1 - flat.astype(np.float64))
    assert np.median(diff[moved > 1e-4]) < 1e-6
    assert np.allclose(s_std.cpu().numpy(), s_gen.cpu().numpy(), rtol=2e-3, atol=1e-3), (s_std, s_gen)
    m1, v1, t1 = p_std.adam_state()
    m2, v2, t2 = p_gen.adam_state()
    assert t1 == t2 == steps
    assert torch.allclose(m1, m2, rtol=1e-3, atol=1e-6) and torch.allclose(v1, v2, rtol=1e-3, atol=1e-9)
    # data-parallel form of the same instantiation: bit-identical to the fused pass
    for _ in range(2):
        for mb in range(N // MB):
            g, _ = p_dp.ppo_minibatch_grad(mb, o, a, adv, ret, b, olp, ov, ol, minibatch=MB, **kw)
            p_dp.adam_step(g, lr=1e-3)
    assert torch.equal(p_dp.params(), p_std.params())
    m3, v3, t3 = p_dp.adam_state()
    assert torch.equal(m3, m1) and torch.equal(v3, v1) and t3 == t1


def test_ppo_pass_is_deterministic_over_many_minibatches():
    """The persistent pass hands activations and parameters between workgroups through two grid barriers per minibatch
    (cross-XCD L2 write-back / L1 invalidate).  A stale read would be timing dependent, so: two handles, identical state,
    160 minibatches with other work queued in between - parameters, Adam moments' effect and statistics must be bit-identical,
    and must also equal a third run made while another stream keeps the chip busy."""
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    rs = np.random.RandomState(3)
    N, MB = 160 * 256, 256
    obs, mask, bits = _data(N, rs)
    flat = init_policy_params(seed=4)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    o, b = t(obs), torch.from_numpy(bits).cuda()
    ref = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
    a, lp, v, ent, lg = ref.act(o, mask_bits=b, seed=5, step=0, want_logits=True)
    adv, ret = t(rs.randn(N)), t(rs.randn(N) * 20 + 50)
    outs = []
    for run in range(3):
        pol = DevicePolicy(256, 64, 284, max_rows=N, params=flat)
        side = torch.cuda.Stream()
        if run == 2:                                          # uneven load: a second stream hammers HBM during the pass
            big = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
            with torch.cuda.stream(side):
                for _ in range(40):
                    big.mul_(1.0001)
        stats = None
        for _ in range(2):                                     # two consecutive passes (Adam step counter carries over)
            stats = pol.ppo_epoch(o, a, adv, ret, b, lp, v, lg, minibatch=MB, lr=3e-4)
        torch.cuda.synchronize()
        outs.append((pol.params().clone(), stats.clone()))
    for k in (1, 2):
        assert torch.equal(outs[0][0], outs[k][0]), 'run %d: parameters differ' % k
        assert torch.equal(outs[0][1], outs[k][1])
    assert torch.isfinite(outs[0][0]).all() and not torch.equal(outs[0][0].cpu(), torch.from_numpy(flat))


def test_training_loop_runs_and_improves_masked_policy(tmp_path):
    """A2C / PPO iterations over the GPU env: runs, stays finite, never plays a masked action."""
    import torch
    import os
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from rl4rs_amd.train import Trainer
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=4)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(300, seed=2, hash_size=2000, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    cfg = {"maxlen": 64, "batch_size": 64, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": 9, "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'),
           "iteminfo_file": os.path.join(d, 'c.csv'), "cache_size": 256, "model_seed": 3, "return_tensors": True}
    for algo in ('A2C', 'PPO'):
        env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(dict(cfg), state_cls=SlateState))
        tr = Trainer(env, algo=algo, seed=1, lr=1e-3, minibatch=128)
        p0 = tr.policy.params().clone()
        outs = [tr.train_iteration() for _ in range(3)]
        assert all(np.isfinite(list(o.values())).all() for o in outs)
        assert not torch.equal(p0, tr.policy.params())
        # the masked policy only produces legal slates: rewards are not zeroed by the violation rule
        assert env.samples.get_violation().all()
        assert outs[-1]['episode_reward_mean'] > 0


def test_wider_hidden_layer_takes_the_same_paths():
    """hidden = 128 (4 hidden tiles, 2-way K split, 81 gradient tasks > 64 wave groups): sampling, both losses and the
    persistent PPO pass against the float64 oracle / the per-minibatch sequence."""
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    from oracle import policy as OP
    HID, A = 128, 284
    rs = np.random.RandomState(21)
    N, MB = 1027, 256
    obs, mask, bits = _data(N, rs)
    n_par = 256 * HID + HID + HID * (A + 1) + A + 1
    flat = init_policy_params(hidden=HID, seed=6) + (rs.randn(n_par) * 0.05).astype(np.float32)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    o, b = t(obs), torch.from_numpy(bits).cuda()
    pol = DevicePolicy(256, HID, A, max_rows=N, params=flat)
    assert pol.n_params == n_par
    a, lp, v, ent, lg = pol.act(o, b, seed=3, step=1, want_logits=True)
    logits, value = OP.forward(flat, obs, mask, hid=HID)
    ok = mask > 0
    assert np.abs(lg.cpu().numpy()[ok] - logits[ok]).max() < 2e-5
    assert np.abs(v.cpu().numpy() - value).max() < 2e-5
    a_np = a.cpu().numpy()
    assert mask[np.arange(N), a_np].all()
    lsm = OP.log_softmax(logits)
    assert np.abs(lp.cpu().numpy() - lsm[np.arange(N), a_np]).max() < 3e-5
    adv, ret = rs.randn(N) * 3, rs.randn(N) * 50 + 100
    old_logp = lsm[np.arange(N), a_np]
    kw = dict(vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=30.0, kl_coeff=0.2)
    ol = np.maximum(logits, -3.4e38).astype(np.float32)
    for algo in (0, 1):
        g, stats = pol.loss_grad(algo, o, a, t(adv), t(ret), mask_bits=b, old_logp=t(old_logp), old_value=t(value), old_logits=t(ol), **kw)
        g_ref, s_ref = OP.loss_and_grad(algo, flat, obs, mask, a_np, adv, ret, old_logp, value, logits, hid=HID, **kw)
        assert np.abs(g.cpu().numpy() - g_ref).max() < 2e-4 * np.abs(g_ref).max(), algo
        assert np.allclose(stats.cpu().numpy(), s_ref, rtol=2e-4, atol=1e-3)
    # persistent pass == minibatch sequence (numerically: see test_ppo_epoch_equals_minibatch_sequence)
    p1 = DevicePolicy(256, HID, A, max_rows=N, params=flat)
    p2 = DevicePolicy(256, HID, A, max_rows=N, params=flat)
    kw2 = dict(vf_coeff=0.5, ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=0.2)
    for lo in range(0, N - MB + 1, MB):
        hi = lo + MB
        g, _ = p1.loss_grad(1, o[lo:hi], a[lo:hi], t(adv)[lo:hi], t(ret)[lo:hi], mask_bits=b[lo:hi], old_logp=t(old_logp)[lo:hi],
                            old_value=t(value)[lo:hi], old_logits=t(ol)[lo:hi], **kw2)
        p1.adam_step(g, lr=1e-3)
    p2.ppo_epoch(o, a, t(adv), t(ret), b, t(old_logp), t(value), t(ol), minibatch=MB, lr=1e-3, **kw2)
    w1, w2 = p1.params().cpu().numpy().astype(np.float64), p2.params().cpu().numpy().astype(np.float64)
    diff = np.abs(w1 - w2)
    assert (diff < 2e-5).mean() > 0.999 and diff.max() <= 2 * (N // MB) * 1e-3, ((diff < 2e-5).mean(), diff.max())


def test_packed_obs_mask_equals_dense_mask(tmp_path):
    """rl4rs_env_obs_mask dtype 4 (packed words, what the trainer hands to the policy) == the dense mask packed on the host,
    along a whole episode."""
    import os
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    d = str(tmp_path)
    text = synth.make_catalog_text(seed=4)
    synth.write_text(os.path.join(d, 'c.csv'), text)
    recs = synth.make_records(70, seed=2, hash_size=2000, special_ids=synth.special_ids_from_text(text))
    synth.write_records(os.path.join(d, 'log.csv'), recs)
    cfg = {"maxlen": 64, "batch_size": 64, "action_size": 284, "class_num": 2, "dense_feature_num": 432, "category_feature_num": 21,
           "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9, "hidden_units": 128, "max_steps": 9,
           "action_emb_size": 32, "sample_file": os.path.join(d, 'log.csv'), "iteminfo_file": os.path.join(d, 'c.csv'),
           "is_eval": True, "cache_size": 64, "return_tensors": True}
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    env.reset()
    for t in range(9):
        live = env.samples._live()
        dense = live.obs_mask(torch.uint8).cpu().numpy().astype(np.uint32)
        want = np.zeros((64, 9), np.uint32)
        for k in range(284):
            want[:, k >> 5] |= dense[:, k] << np.uint32(k & 31)
        got = live.obs_mask_bits().cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), t
        env.step(env.offline_action)


def test_rawstate_policy_forward_matches_oracle(tmp_path):
    """rl4rs_rawpolicy_* (rllib_rawstate_model.py + mask wrapper) against the numpy fp64 restatement, then driven by the
    raw features of a rawstate_as_obs env."""
    import torch
    from rl4rs_amd.device import DeviceRawPolicy
    from rl4rs_amd.nets.rawpolicy import init_rawpolicy_weights
    from oracle import policy as OP
    cfg = {"maxlen": 64, "action_size": 284, "dense_feature_num": 432, "category_feature_num": 21,
           "category_hash_size": 3000, "seq_num": 2, "emb_size": 128, "hidden_units": 128}
    rs = np.random.RandomState(4)
    N = 300
    w = init_rawpolicy_weights(cfg, seed=2, emb_scale=0.5, head_std=1.0, bias_noise=0.2)
    cat = rs.randint(0, 3000, size=(N, 21)).astype(np.int32)
    dense = np.abs(rs.randn(N, 432) * 2).astype(np.float32)
    seqs = [rs.randint(0, 284, size=(N, 64)).astype(np.int32) for _ in range(2)]
    seqs[1][::2] = 0
    _, mask, bits = _data(N, rs)
    pol = DeviceRawPolicy(cfg, w, max_rows=N)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a, lp, v, ent, lg = pol.act(t(cat), t(dense), [t(s) for s in seqs], t(bits), seed=5, step=3, want_logits=True)
    logits_ref, value_ref = OP.rawstate_forward(w, cat, dense, seqs, mask)
    lsm = OP.log_softmax(logits_ref)
    a_np = a.cpu().numpy()
    assert mask[np.arange(N), a_np].all()                       # only allowed actions are drawn
    ok = mask.astype(bool)
    assert np.abs(lg.cpu().numpy()[ok] - logits_ref[ok]).max() < 2e-4
    assert np.abs(v.cpu().numpy() - value_ref).max() < 2e-4
    assert np.abs(lp.cpu().numpy() - lsm[np.arange(N), a_np]).max() < 2e-4
    p = np.exp(lsm)
    ent_ref = -(np.where(p > 0, p * lsm, 0.0)).sum(axis=1)
    assert np.abs(ent.cpu().numpy() - ent_ref).max() < 2e-4
    # evaluate() of the drawn actions reproduces act(); no mask = plain logits
    lp2, v2, ent2, _ = pol.evaluate(t(cat), t(dense), [t(s) for s in seqs], a, t(bits))
    assert torch.equal(lp, lp2) and torch.equal(v, v2) and torch.equal(ent, ent2)
    _, _, _, lg0 = pol.evaluate(t(cat), t(dense), [t(s) for s in seqs], a, None, want_logits=True)
    assert np.abs(lg0.cpu().numpy() - OP.rawstate_forward(w, cat, dense, seqs, None)[0]).max() < 2e-4
    # sampling follows the masked categorical
    rep = 20000
    one = lambda x: t(np.repeat(x[:1], rep, axis=0))
    pol2 = DeviceRawPolicy(cfg, w, max_rows=rep)
    aa = pol2.act(one(cat), one(dense), [one(s) for s in seqs], one(bits), seed=11, step=0)[0].cpu().numpy()
    freq = np.bincount(aa, minlength=284) / float(rep)
    assert np.abs(freq - p[0]).max() < 0.02
    with pytest.raises(Exception, match='max_rows'):
        pol.act(one(cat), one(dense), [one(s) for s in seqs], one(bits))
    pol.close()
    pol2.close()


def test_rawstate_policy_drives_rawstate_env(tmp_path):
    """config['rawstate_as_obs'] + return_tensors: the env hands out device tensors of the raw features and the raw-state
    policy acts on them; compared per step with the oracle forward on the oracle env's raw state."""
    import os
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.device import DeviceRawPolicy
    from rl4rs_amd.nets.rawpolicy import init_rawpolicy_weights
    from rl4rs.env.slate import SlateRecEnv, SlateState
    from oracle import policy as OP
    B, T = 16, 9
    d = str(tmp_path)
    cat_path, log_path = os.path.join(d, 'item_info.csv'), os.path.join(d, 'log.csv')
    cat_text = synth.make_catalog_text(seed=21)
    synth.write_text(cat_path, cat_text)
    records = synth.make_records(B, pages=1, seed=8, illegal_frac=0.0, hash_size=5000,
                                 special_ids=synth.special_ids_from_text(cat_text))
    synth.write_records(log_path, records)
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 5000, "seq_num": 2, "emb_size": 128,
           "page_items": 9, "hidden_units": 128, "max_steps": T, "action_emb_size": 32,
           "sample_file": log_path, "iteminfo_file": cat_path, "is_eval": True, "cache_size": B, "model_seed": 3,
           "support_rllib_mask": True, "rawstate_as_obs": True, "return_tensors": True}
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    w = init_rawpolicy_weights(cfg, seed=1, emb_scale=0.5, head_std=1.0, bias_noise=0.1)
    pol = DeviceRawPolicy(cfg, w, max_rows=B)
    obs = env.reset(reset_file=True)
    total = torch.zeros(B, dtype=torch.float64, device='cuda')
    for t in range(T):
        assert set(obs) == {'category_feature', 'dense_feature', 'sequence_feature', 'action_mask'}
        mask = obs['action_mask']
        mask_np = mask.cpu().numpy()
        W = (284 + 31) // 32
        bits = np.zeros((B, W * 32), dtype=np.int64)
        bits[:, :284] = mask_np
        bits = (bits.reshape(B, W, 32) << np.arange(32)).sum(axis=2).astype(np.uint32).view(np.int32)
        a, lp, v, ent, lg = pol.act(obs['category_feature'], obs['dense_feature'], obs['sequence_feature'],
                                    torch.from_numpy(bits).cuda(), seed=7, step=t, want_logits=True)
        logits_ref, value_ref = OP.rawstate_forward(w, obs['category_feature'].cpu().numpy(), obs['dense_feature'].cpu().numpy(),
                                                    [s.cpu().numpy() for s in obs['sequence_feature']], mask_np)
        ok = mask_np.astype(bool)
        assert np.abs(lg.cpu().numpy()[ok] - logits_ref[ok]).max() < 2e-4
        assert np.abs(v.cpu().numpy() - value_ref).max() < 2e-4
        assert mask_np[np.arange(B), a.cpu().numpy()].all()
        obs, reward, done, info = env.step(a)
        total += reward
    assert bool(done[0]) if not torch.is_tensor(done) else bool(done.all())
    assert float(total.abs().sum()) > 0


@pytest.mark.parametrize('algo', [0, 1])
def test_rawstate_policy_loss_gradients_match_autograd(algo):
    """rl4rs_rawtrain_loss_grad (A2C / PPO on the raw-state policy) against torch float64 autograd, then one Adam step."""
    import torch
    from rl4rs_amd.device import DeviceRawTrainer
    from rl4rs_amd.nets.rawpolicy import init_rawpolicy_weights
    from oracle import policy as OP
    cfg = {"maxlen": 64, "action_size": 284, "dense_feature_num": 432, "category_feature_num": 21,
           "category_hash_size": 3000, "seq_num": 2, "emb_size": 128, "hidden_units": 128}
    rs = np.random.RandomState(algo + 3)
    N = 700
    w = init_rawpolicy_weights(cfg, seed=2, emb_scale=0.5, head_std=1.0, bias_noise=0.2)
    cat = rs.randint(0, 3000, size=(N, 21)).astype(np.int32)
    dense = np.abs(rs.randn(N, 432) * 2).astype(np.float32)
    seqs = [rs.randint(0, 284, size=(N, 64)).astype(np.int32) for _ in range(2)]
    _, mask, bits = _data(N, rs)
    old_w = dict((k, v + (rs.randn(*v.shape) * 0.01).astype(np.float32)) for k, v in w.items())
    old_logits, old_value = OP.rawstate_forward(old_w, cat, dense, seqs, mask)
    old_lsm = OP.log_softmax(old_logits)
    actions = np.array([rs.choice(np.nonzero(mask[i])[0]) for i in range(N)])
    old_logp = old_lsm[np.arange(N), actions]
    adv = rs.randn(N) * 3
    ret = rs.randn(N) * 50 + 100
    kw = dict(vf_coeff=0.5, ent_coeff=0.01, clip=0.3, vf_clip=30.0, kl_coeff=0.2)
    pol = DeviceRawTrainer(cfg, w, max_rows=N)
    t = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).cuda()
    dseqs = [t(q) for q in seqs]
    stats = pol.loss_grad(algo, t(cat), t(dense), dseqs, t(actions), t(adv), t(ret), mask_bits=t(bits), old_logp=t(old_logp),
                          old_value=t(old_value), old_logits=t(np.maximum(old_logits, -3.4e38).astype(np.float32)), **kw)
    g = dict((k, v.cpu().numpy()) for k, v in pol.gradients().items())
    g_ref, s_ref = OP.rawstate_loss_and_grad(algo, w, cat, dense, seqs, mask, actions, adv, ret, old_logp, old_value, old_logits, **kw)
    assert set(g) == set(g_ref)
    for k in sorted(g_ref):
        scale = np.abs(g_ref[k]).max()
        assert np.abs(g[k] - g_ref[k]).max() < 3e-4 * max(scale, 1e-8), (k, np.abs(g[k] - g_ref[k]).max(), scale)
    assert np.allclose(stats.cpu().numpy(), s_ref, rtol=2e-4, atol=1e-3)
    # the trainer acts like the forward-only handle built from the same weights
    a1 = pol.act(t(cat), t(dense), dseqs, t(bits), seed=4, step=1)[0]
    from rl4rs_amd.device import DeviceRawPolicy
    ref = DeviceRawPolicy(cfg, w, max_rows=N)
    a2 = ref.act(t(cat), t(dense), dseqs, t(bits), seed=4, step=1)[0]
    assert (a1 == a2).float().mean().item() > 0.995          # same draws up to fp32 summation-order ties
    before = pol.weights()['ctx_w'].clone()
    pol.adam_step(lr=1e-3, grad_clip=10.0)
    assert not torch.equal(before, pol.weights()['ctx_w'])
    pol.close()
    ref.close()


@pytest.mark.parametrize('algo', ['A2C', 'PPO'])
def test_rawstate_training_loop(tmp_path, algo):
    """RawStateTrainer: rollouts of the raw-state policy on a rawstate_as_obs env + A2C / PPO updates on the device."""
    import os
    import torch
    import rl4rs_amd
    from rl4rs_amd import synth
    from rl4rs_amd.train import RawStateTrainer
    from rl4rs.env.slate import SlateRecEnv, SlateState
    B, T = 64, 9
    d = str(tmp_path)
    cat_path, log_path = os.path.join(d, 'item_info.csv'), os.path.join(d, 'log.csv')
    cat_text = synth.make_catalog_text(seed=21)
    synth.write_text(cat_path, cat_text)
    synth.write_records(log_path, synth.make_records(B + 3, pages=1, seed=8, illegal_frac=0.0, hash_size=5000,
                                                     special_ids=synth.special_ids_from_text(cat_text)))
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 5000, "seq_num": 2, "emb_size": 128,
           "page_items": 9, "hidden_units": 128, "max_steps": T, "action_emb_size": 32,
           "sample_file": log_path, "iteminfo_file": cat_path, "is_eval": False, "cache_size": B, "model_seed": 3,
           "support_rllib_mask": True, "rawstate_as_obs": True, "return_tensors": True}
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    env.seed(5)
    tr = RawStateTrainer(env, algo=algo, seed=1, lr=1e-3, minibatch=128)
    before = tr.policy.weights()['ctx_w'].clone()
    outs = [tr.train_iteration() for _ in range(3)]
    assert all(np.isfinite(list(o.values())).all() for o in outs)
    assert outs[-1]['iteration'] == 3 and outs[0]['entropy'] > 0
    assert not torch.equal(before, tr.policy.weights()['ctx_w'])
    # every sampled action respected the env's mask: no violation-zeroed slates from illegal picks of the policy itself
    assert outs[-1]['episode_reward_mean'] > 0
