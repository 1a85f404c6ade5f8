"""The continuous-action offline learner of BASELINE configs[4] ('BCQ-conti', script/batchrl_trainer.py:61-73 = d3rlpy.algos.BCQ
on default encoders) against the float64 torch restatement in oracle/offline_conti.py (PARITY UNPINNED: d3rlpy is absent):
the amlp network (forward with shared observations, every parameter gradient, the action-input gradient), the conditional-VAE
loss, the lam-weighted twin target over sampled actions, the critic and actor gradients, the greedy sampled action (integer
pick: identical wherever the fp32 and fp64 values do not tie), whole updates with supplied noise tracked for several steps, an
end-to-end fit on the continuous dataset the device env generates followed by a policy -> K-NN rollout, and two ranks.

Tolerances: forward 2e-4 abs on O(1) outputs (fp32 MFMA GEMMs over K <= 298), gradients 2e-3 relative to the largest entry of
each array, parameters after k Adam steps 2e-4 abs (Adam's 1/sqrt(v) amplifies rounding of near-zero gradients)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, E, L = 266, 32, 32


def _close(got, want, rel, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(np.abs(want).max(), 1e-12)
    err = np.abs(got - want).max()
    assert err <= rel * scale, '%s: max err %.3e vs scale %.3e' % (what, err, scale)


def _pair(act_dim, out_dim, seed, head_act='none', max_rows=512, heads=1, obs_dim=D):
    from oracle.offline_conti import OracleAMLP
    from rl4rs_amd import device as Dv
    from rl4rs_amd.offline_rl import init_amlp_params
    p = init_amlp_params(obs_dim, act_dim, out_dim, seed=seed, heads=heads)
    return Dv.DeviceAMLP(obs_dim, act_dim, out_dim, p, head_act=head_act, max_rows=max_rows), OracleAMLP(p, head_act), p


def _batch(n, seed):
    """rows shaped like the d3rl-mode observation (256 floats | 9 ids | step) with unit-norm 32-d actions"""
    rs = np.random.RandomState(seed)
    x = rs.randn(n, D).astype(np.float32)
    x[:, 256:265] = rs.randint(0, 284, size=(n, 9))
    x[:, 265] = rs.randint(0, 10, size=n)
    a = rs.randn(n, E).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    rew = (rs.rand(n) * 5).astype(np.float32)
    ter = (rs.rand(n) < 0.15).astype(np.float32)
    return x, a, rew, ter


@pytest.fixture
def amlp_fused():
    """switches the fused minibatch launches of the amlp networks (amlp_fused.hpp) for one test; on again afterwards"""
    from rl4rs_amd import device as Dv
    yield Dv.amlp_set_fused
    Dv.amlp_set_fused(True)


@pytest.mark.parametrize('fused', [True, False, 2])
@pytest.mark.parametrize('act_dim,out_dim,head_act,rep', [(E, 1, 'none', 1), (E, 1, 'none', 7), (L, E, 'tanh', 5), (E, 2 * L, 'none', 1),
                                                          (0, 2 * E, 'none', 1), (E, E, 'tanh', 1)])
def test_amlp_forward_and_gradients(amlp_fused, fused, act_dim, out_dim, head_act, rep):
    """rep = 1 cases: the fused one-launch forward / three-launch backward (default) and the per-layer launches, same bars;
    R = 36 and 255 rows (whole and ragged 4-row workgroups of the fused form)"""
    if fused == 2 and rep != 1:
        pytest.skip('the 8-row fused form only differs for rep = 1 calls')
    amlp_fused(fused)
    for R in (36, 255) if rep == 1 else (36,):
        _check_amlp_forward_and_gradients(act_dim, out_dim, head_act, rep, R)


def _check_amlp_forward_and_gradients(act_dim, out_dim, head_act, rep, R):
    import torch
    dev, orc, _ = _pair(act_dim, out_dim, 11, head_act, heads=2 if out_dim == 2 * L and act_dim else 1)
    N = R * rep
    rs = np.random.RandomState(5)
    x = _batch(R, 6)[0]
    a = rs.randn(N, act_dim).astype(np.float32) * 0.5 if act_dim else None
    xd = torch.from_numpy(x).cuda()
    ad = torch.from_numpy(a).cuda() if act_dim else None
    out = dev.forward(xd, ad, rep=rep)
    xr = np.repeat(x, rep, axis=0)
    at = torch.tensor(a, dtype=torch.float64, requires_grad=True) if act_dim else None
    want = orc(xr, at)
    assert np.abs(out.cpu().numpy() - want.detach().numpy()).max() < 2e-4 * max(1.0, float(want.detach().abs().max()))
    # gradient of sum(out * w): the device takes the gradient wrt the PRE-activation head output
    w = rs.randn(N, out_dim).astype(np.float32)
    (want * torch.from_numpy(w).double()).sum().backward()
    o = out.cpu().numpy()
    dpre = w * (1.0 - o * o) if head_act == 'tanh' else w
    dact = dev.backward(xd, ad, torch.from_numpy(np.ascontiguousarray(dpre, np.float32)).cuda(), rep=rep, want_dact=bool(act_dim))
    g, gw = dev.gradients(), orc.grads()
    for k in gw:
        _close(g[k].cpu().numpy(), gw[k], 2e-3, 'grad %s' % k)
    if act_dim:
        _close(dact.cpu().numpy(), at.grad.numpy(), 2e-3, 'dact')
    # input gradient only: the parameter gradients in the handle stay as they are
    before = dev.flat_gradient().clone()
    dev.forward(xd, ad, rep=rep)
    dev.backward(xd, ad, torch.from_numpy(np.ascontiguousarray(dpre * 2, np.float32)).cuda(), rep=rep, want_dact=bool(act_dim), want_param_grad=False)
    assert torch.equal(before, dev.flat_gradient())
    dev.close()


@pytest.mark.parametrize('fused', [True, False])
def test_twin_networks_in_one_launch_equal_single_calls(amlp_fused, fused):
    """rl4rs_amlp_forward_multi / _backward_multi (the twin critics over the same rows as one launch each way) == one
    rl4rs_amlp_forward / _backward per network, bit for bit: outputs, action-input gradients, parameter gradients"""
    import torch
    from rl4rs_amd import device as Dv
    amlp_fused(fused)
    rs = np.random.RandomState(12)
    N = 203
    x = torch.from_numpy(_batch(N, 13)[0]).cuda()
    a = torch.from_numpy(rs.randn(N, E).astype(np.float32) * 0.5).cuda()
    pair = [_pair(E, 1, 31)[0], _pair(E, 1, 32)[0]]
    solo = [_pair(E, 1, 31)[0], _pair(E, 1, 32)[0]]
    outs = Dv.amlp_forward_multi(pair, x, a)
    douts = [torch.from_numpy(rs.randn(N, 1).astype(np.float32)).cuda() for _ in range(2)]
    dacts = Dv.amlp_backward_multi(pair, x, a, douts, want_dact=True)
    for net, o, d, da in zip(solo, outs, douts, dacts):
        assert torch.equal(net.forward(x, a), o)
        assert torch.equal(net.backward(x, a, d, want_dact=True), da)
    for p, q in zip(pair, solo):
        assert torch.equal(p.flat_gradient(), q.flat_gradient())
        p.close()
        q.close()


@pytest.mark.parametrize('nograd', ['fp32', 'fp16x2'])
def test_update_as_one_library_call_equals_the_per_phase_calls(nograd):
    """rl4rs_bcq_update (the whole update as one host call, the default on one rank) issues the same launches with the same
    arguments as BCQ.update's per-phase path: parameters and Adam state bit-identical over four updates with shared noise,
    with and without the actor phase (update_actor_interval = 2)"""
    import torch
    B, n = 64, 8
    a, _ = _learner_pair(61, B, n, nograd=nograd)
    b, _ = _learner_pair(61, B, n, nograd=nograd)
    a.update_actor_interval = b.update_actor_interval = 2
    assert a.one_call
    b.one_call = False
    rs = np.random.RandomState(62)
    f = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32))
    for it in range(4):
        x, act, rew, ter = _batch(B, 70 + it)
        nx = _batch(B, 80 + it)[0]
        noise = dict(eps=f(rs.randn(B, L)), z_target=f(rs.randn(B * n, L)), z_actor=f(rs.randn(B, L)))
        keep = dict((k, v.clone()) for k, v in noise.items())
        args = [f(v).cuda() for v in (x, act, rew, nx, ter)]
        ma = a.update(*args, noise=noise)
        mb = b.update(*args, noise=noise)
        assert set(ma) == set(mb) and ('actor_loss' in ma) == (it % 2 == 0)
        for k in ma:
            # (the reported losses are reductions in another order - one metrics kernel against torch.dot / sum: equal to rounding)
            assert abs(float(ma[k]) - float(mb[k])) <= 2e-6 * max(1.0, abs(float(mb[k]))), (it, k, float(ma[k]), float(mb[k]))
        for k in noise:
            assert torch.equal(noise[k], keep[k])                 # the caller's noise is never modified
    for na, nb in zip(a.nets, b.nets):
        assert torch.equal(na.flat_params(), nb.flat_params())
    for na, nb in zip((a.imit_enc, a.imit_dec, a.policy, a.q1, a.q2), (b.imit_enc, b.imit_dec, b.policy, b.q1, b.q2)):
        (ma_, va_, ta_), (mb_, vb_, tb_) = na.adam_state(), nb.adam_state()
        assert torch.equal(ma_, mb_) and torch.equal(va_, vb_) and ta_ == tb_
    assert a.total_step == b.total_step == 4
    a.close()
    b.close()


def test_one_call_update_rejects_what_the_per_phase_path_rejects():
    """the one-call update takes raw pointers: a host tensor, a float64 reward, a bool terminal or a wrong-width observation must
    raise before anything reaches the library (they used to become a memory fault or silent garbage)"""
    import torch
    B, n = 64, 8
    a, _ = _learner_pair(63, B, n)
    f = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32)).cuda()
    x, act, rew, ter = _batch(B, 71)
    nx = _batch(B, 72)[0]
    good = [f(v) for v in (x, act, rew, nx, ter)]
    bad = [
        [good[0].cpu()] + good[1:],
        good[:2] + [good[2].double()] + good[3:],
        good[:4] + [good[4] > 0.5],
        [good[0][:, :-1]] + good[1:],
        good[:1] + [good[1][:, :-1]] + good[2:],
        good[:2] + [good[2][:-1]] + good[3:],
        good[:3] + [good[3][:-1]] + good[4:],
    ]
    before = [net.flat_params().clone() for net in a.nets]
    for args in bad:
        with pytest.raises(AssertionError):
            a.update(*args)
    assert a.total_step == 0
    for net, p0 in zip(a.nets, before):
        assert torch.equal(net.flat_params(), p0)
    a.update(*good)                                # and the well-formed batch still goes through
    a.update(good[0], good[1], good[2].reshape(B, 1), good[3], good[4].reshape(B, 1))     # [B, 1] columns are the same memory
    assert a.total_step == 2
    a.close()


def test_adam_multi_equals_separate_launches():
    """rl4rs_amlp_adam_multi (Adam of several networks + soft target updates as one launch) == rl4rs_amlp_adam_step and
    rl4rs_amlp_soft_update per network, bit for bit, over three steps; a soft-update-only entry leaves its source untouched"""
    import torch
    from rl4rs_amd import device as Dv
    rs = np.random.RandomState(9)

    def nets(seed):
        a = _pair(E, 1, seed)[0]
        b = _pair(L, E, seed + 1, 'tanh')[0]
        ta = _pair(E, 1, seed + 2)[0]
        tb = _pair(E, 1, seed + 3)[0]
        return a, b, ta, tb

    one, two = nets(21), nets(21)
    for step in range(3):
        ga = torch.from_numpy(rs.randn(one[0].n_params).astype(np.float32)).cuda()
        gb = torch.from_numpy(rs.randn(one[1].n_params).astype(np.float32)).cuda()
        for grp in (one, two):
            grp[0].set_flat_gradient(ga)
            grp[1].set_flat_gradient(gb)
        # separate launches
        one[0].adam_step(1e-3)
        one[1].adam_step(3e-4)
        one[2].soft_update_from(one[0], 0.005)
        one[3].soft_update_from(one[2], 0.005)             # soft-update-only entry: source one[2] (no step)
        # one launch: [a (step, target ta), b (step), ta (no step, target tb)] - ta is written by entry 0 and read by entry 2 in
        # the SAME launch, so that dependent pair goes in a second call, like the learners order their phases
        Dv.amlp_adam_multi([two[0], two[1]], [1e-3, 3e-4], targets=[two[2], None], tau=0.005)
        Dv.amlp_adam_multi([two[2]], [0.0], targets=[two[3]], tau=0.005, step=[False])
        for x, y in zip(one, two):
            assert torch.equal(x.flat_params(), y.flat_params()), step
        for x, y in zip(one[:2], two[:2]):
            mx, vx, tx = x.adam_state()
            my, vy, ty = y.adam_state()
            assert torch.equal(mx, my) and torch.equal(vx, vy) and tx == ty == step + 1
    assert one[2].adam_state()[2] == 0 and two[2].adam_state()[2] == 0
    for grp in (one, two):
        for net in grp:
            net.close()


@pytest.mark.parametrize('obs_dim,act_dim,h1,h2,out_dim,R,rep', [(37, 5, 48, 40, 3, 17, 3), (266, 32, 256, 256, 1, 300, 31), (10, 0, 33, 65, 7, 70, 1),
                                                                 (266, 32, 256, 256, 32, 5000, 1)])
def test_amlp_odd_shapes_and_many_rows(obs_dim, act_dim, h1, h2, out_dim, R, rep):
    """the small-M GEMM forms behind the amlp (k_gemm_small with one or two operand pairs, k_gemm_nt, k_gemm_tn + bias sums) on
    widths that are multiples of nothing (scalar-load paths, k / row / column tails), on the 31-rows-per-observation layout of
    the CQL critic pass (chunked sample-axis reductions: 9300 rows), and above the row count where the big-tile GEMM takes over"""
    import torch
    from oracle.offline_conti import OracleAMLP
    from rl4rs_amd import device as Dv
    from rl4rs_amd.offline_rl import init_amlp_params
    p = init_amlp_params(obs_dim, act_dim, out_dim, hidden1=h1, hidden2=h2, seed=7)
    N = R * rep
    dev = Dv.DeviceAMLP(obs_dim, act_dim, out_dim, p, hidden1=h1, hidden2=h2, max_rows=N)
    orc = OracleAMLP(p)
    rs = np.random.RandomState(3)
    x = rs.randn(R, obs_dim).astype(np.float32)
    a = rs.randn(N, act_dim).astype(np.float32) if act_dim else None
    xd = torch.from_numpy(x).cuda()
    ad = torch.from_numpy(a).cuda() if act_dim else None
    out = dev.forward(xd, ad, rep=rep)
    at = torch.tensor(a, dtype=torch.float64, requires_grad=True) if act_dim else None
    want = orc(np.repeat(x, rep, axis=0), at)
    assert np.abs(out.cpu().numpy() - want.detach().numpy()).max() < 2e-4 * max(1.0, float(want.detach().abs().max()))
    w = (rs.randn(N, out_dim) / N).astype(np.float32)
    (want * torch.from_numpy(w).double()).sum().backward()
    dact = dev.backward(xd, ad, torch.from_numpy(w).cuda(), rep=rep, want_dact=bool(act_dim))
    g, gw = dev.gradients(), orc.grads()
    for k in gw:
        _close(g[k].cpu().numpy(), gw[k], 2e-3, 'grad %s' % k)
    if act_dim:
        _close(dact.cpu().numpy(), at.grad.numpy(), 2e-3, 'dact')
    dev.close()


def test_backward_must_follow_the_forward_of_the_same_rows():
    import torch
    from rl4rs_amd._lib import Rl4rsHipError
    dev, _, _ = _pair(E, 1, 3, max_rows=64)
    x, a, _, _ = _batch(64, 1)
    xd, ad = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
    dev.forward(xd, ad)
    with pytest.raises(Rl4rsHipError):
        dev.backward(xd[:32].contiguous(), ad[:32].contiguous(), torch.zeros(32, 1, device='cuda'))
    with pytest.raises(Rl4rsHipError):
        dev.forward(torch.zeros(65, D, device='cuda'), torch.zeros(65, E, device='cuda'))


def test_cvae_loss_and_gradients():
    import torch
    from oracle import offline_conti as O
    from rl4rs_amd import device as Dv
    enc, oenc, _ = _pair(E, 2 * L, 21, heads=2)
    dec, odec, _ = _pair(L, E, 22, 'tanh')
    x, a, _, _ = _batch(256, 23)
    eps = np.random.RandomState(24).randn(256, L).astype(np.float32)
    xd, ad, ed = [torch.from_numpy(v).cuda() for v in (x, a, eps)]
    for beta in (0.5, 0.0):
        enc_out = enc.forward(xd, ad)
        z = Dv.cvae_sample(enc_out, ed)
        y = dec.forward(xd, z)
        loss2, d_dec = Dv.cvae_loss(y, ad, enc_out)
        dz = dec.backward(xd, z, d_dec, want_dact=True)
        enc.backward(xd, ad, Dv.cvae_encoder_grad(enc_out, ed, dz, beta))
        oenc.zero_grad()
        odec.zero_grad()
        want = O.cvae_loss(oenc, odec, x, a, eps, beta)
        want.backward()
        got = float(loss2[0] / E + beta * loss2[1] / L)
        assert abs(got - float(want.detach())) < 1e-4 * max(1.0, abs(float(want.detach()))), (got, float(want.detach()))
        for dev, orc, name in ((enc, oenc, 'encoder'), (dec, odec, 'decoder')):
            g, gw = dev.gradients(), orc.grads()
            for k in gw:
                _close(g[k].cpu().numpy(), gw[k], 2e-3, 'beta %.1f %s grad %s' % (beta, name, k))


def test_logstd_clamp_blocks_the_gradient():
    """torch.clamp(logstd, -20, 2): outside the range sigma is the bound and no gradient reaches the logstd head."""
    import torch
    from rl4rs_amd import device as Dv
    enc_out = torch.zeros(4, 2 * L, device='cuda')
    enc_out[:, L:] = torch.tensor([3.0, -25.0, 1.0, 2.0], device='cuda')[:, None]
    eps = torch.ones(4, L, device='cuda')
    z = Dv.cvae_sample(enc_out, eps).cpu().numpy()
    assert np.allclose(z[:, 0], np.exp([2.0, -20.0, 1.0, 2.0]), rtol=1e-6)
    d = Dv.cvae_encoder_grad(enc_out, eps, torch.ones(4, L, device='cuda'), 0.5).cpu().numpy()
    assert (d[0, L:] == 0).all() and (d[1, L:] == 0).all() and (d[2, L:] != 0).all() and (d[3, L:] != 0).all()


def _learner_pair(seed, B, n, scale=None, nograd='fp32'):
    """a device BCQ learner and float64 oracle networks holding the same parameters (``scale``: {network: factor} applied to
    its initial parameters first).  ``nograd='fp16x2'``: the sampled-action forwards through the fused fp16x2 kernel at ANY row
    count (by default it takes over from 4096 rows: the bench's 25 600 / 409 600, not a test's few hundred)"""
    from oracle.offline_conti import OracleAMLP
    from rl4rs_amd.offline_rl import BCQ
    bcq = BCQ({'action_emb_size': E}, D, batch_size=B, n_action_samples=n, predict_rows=64, seed=seed, nograd_precision=nograd)
    if nograd == 'fp16x2':
        for net in bcq.nets:
            assert net.h16_ok
            net.H16_MIN_ROWS = 0
    for k, f in (scale or {}).items():
        net = getattr(bcq, k)
        net.set_flat_params((net.flat_params() * f).contiguous())
    names = ('imit_enc', 'imit_dec', 'policy', 'policy_targ', 'q1', 'q2', 'q1_targ', 'q2_targ')
    heads = dict(imit_dec='tanh', policy='tanh', policy_targ='tanh')
    orc = dict((k, OracleAMLP(dict((pk, pv.cpu().numpy()) for pk, pv in getattr(bcq, k).weights().items()), heads.get(k, 'none')))
               for k in names)
    return bcq, orc


def test_target_critic_and_actor_gradients():
    import torch
    from oracle import offline_conti as O
    from rl4rs_amd import device as Dv
    B, n = 64, 10
    bcq, orc = _learner_pair(31, B, n)
    # decorrelate the targets from the online networks so the test sees which one is used where
    for k in ('policy_targ', 'q1_targ', 'q2_targ'):
        net = getattr(bcq, k)
        net.set_flat_params((net.flat_params() * 1.3).contiguous())
        orc[k] = O.OracleAMLP(dict((pk, pv.cpu().numpy()) for pk, pv in net.weights().items()), orc[k].head_act)
    x, a, rew, ter = _batch(B, 32)
    nx = _batch(B, 33)[0]
    nx[ter > 0.5] = 0.0
    rs = np.random.RandomState(34)
    zt = rs.randn(B * n, L).astype(np.float32)
    za = rs.randn(B, L).astype(np.float32)
    xd, ad, nd, rd, td = [torch.from_numpy(v).cuda() for v in (x, a, nx, rew, ter)]
    # target
    _, _, a_next = bcq._sample_actions(nd, torch.from_numpy(zt).cuda().clamp(-0.5, 0.5), bcq.policy_targ, n)
    q1n = bcq.q1_targ.forward(nd, a_next, rep=n)
    q2n = bcq.q2_targ.forward(nd, a_next, rep=n)
    y, _ = Dv.bcq_target(q1n, q2n, n, bcq.lam, rd, td, bcq.gamma)
    want_y = O.bcq_target(orc['imit_dec'], orc['policy_targ'], [orc['q1_targ'], orc['q2_targ']], nx, zt, n, bcq.scale, bcq.lam, rew, ter, bcq.gamma)
    assert np.abs(y.cpu().numpy() - want_y.numpy()).max() < 5e-4 * max(1.0, float(want_y.abs().max()))
    # critic
    q1v, q2v = bcq.q1.forward(xd, ad), bcq.q2.forward(xd, ad)
    loss2, dq1, dq2 = Dv.critic_mse(q1v, q2v, y)
    bcq.q1.backward(xd, ad, dq1)
    bcq.q2.backward(xd, ad, dq2)
    want = O.critic_loss([orc['q1'], orc['q2']], x, a, want_y)
    want.backward()
    assert abs(float(loss2.sum()) - float(want.detach())) < 1e-3 * max(1.0, abs(float(want.detach())))
    for k in ('q1', 'q2'):
        g, gw = getattr(bcq, k).gradients(), orc[k].grads()
        for pk in gw:
            _close(g[pk].cpu().numpy(), gw[pk], 2e-3, 'critic %s grad %s' % (k, pk))
    # actor: gradient reaches the policy only
    sampled, t, a_pi = bcq._sample_actions(xd, torch.from_numpy(za).cuda().clamp(-0.5, 0.5), bcq.policy, 1)
    qv = bcq.q1.forward(xd, a_pi)
    da = bcq.q1.backward(xd, a_pi, torch.full((B, 1), -1.0 / B, device='cuda'), want_dact=True, want_param_grad=False)
    bcq.policy.backward(xd, sampled, Dv.residual_grad(sampled, t, bcq.scale, da))
    for v in orc.values():
        v.zero_grad()
    want = O.actor_loss(orc['imit_dec'], orc['policy'], orc['q1'], x, za, bcq.scale)
    want.backward()
    assert abs(float(-qv.mean()) - float(want.detach())) < 2e-4 * max(1.0, abs(float(want.detach())))
    g, gw = bcq.policy.gradients(), orc['policy'].grads()
    for pk in gw:
        _close(g[pk].cpu().numpy(), gw[pk], 2e-3, 'actor grad %s' % pk)
    bcq.close()


def test_residual_clamp_and_target_rules():
    """integer / piecewise rules on their own: the clamp of the residual policy stops the gradient, the target takes the FIRST
    maximum of (1 - lam) max + lam min, terminal rows keep only the reward."""
    import torch
    from rl4rs_amd import device as Dv
    a = torch.tensor([[0.99, -0.99, 0.2, 1.0]], device='cuda')
    t = torch.tensor([[1.0, -1.0, 0.5, 0.0]], device='cuda')
    out = Dv.residual_action(a, t, 0.05).cpu().numpy()
    assert np.allclose(out, [[1.0, -1.0, 0.225, 1.0]])
    d = Dv.residual_grad(a, t, 0.05, torch.ones_like(a)).cpu().numpy()
    assert d[0, 0] == 0 and d[0, 1] == 0 and np.isclose(d[0, 2], 0.05 * 0.75) and np.isclose(d[0, 3], 0.05)   # clamp passes at the bound
    q1 = torch.tensor([1.0, 4.0, 2.0, 4.0, 0.0, -1.0], device='cuda')
    q2 = torch.tensor([3.0, 0.0, 2.0, 0.0, 5.0, -1.0], device='cuda')
    y, best = Dv.bcq_target(q1, q2, 3, 0.75, torch.tensor([1.0, 2.0], device='cuda'), torch.tensor([0.0, 1.0], device='cuda'), 0.5, want_best=True)
    # row 0: mixes 1.5, 1.0, 2.0 -> 2.0 at j=2; row 1: 1.0, 1.25, -1 -> j=1, terminal
    assert best.cpu().tolist() == [2, 1]
    assert np.allclose(y.cpu().numpy(), [1.0 + 0.5 * 2.0, 2.0])
    v, best = Dv.bcq_target(torch.tensor([2.0, 5.0, 5.0, 1.0], device='cuda'), None, 4, 0.0, want_best=True)
    assert best.cpu().tolist() == [1] and float(v) == 5.0


@pytest.mark.parametrize('nograd', ['fp32', 'fp16x2'])
def test_updates_track_the_fp64_restatement(nograd):
    """three whole updates (imitator, critic, actor, soft target updates) with the same noise on both sides"""
    import torch
    from oracle import offline_conti as O
    from oracle.offline_rl import torch_adam
    B, n, steps = 64, 8, 3
    bcq, orc = _learner_pair(41, B, n, nograd=nograd)
    P = dict((k, v.numpy_params()) for k, v in orc.items())
    heads = dict((k, v.head_act) for k, v in orc.items())
    M = dict((k, dict((pk, np.zeros_like(pv)) for pk, pv in P[k].items())) for k in P)
    V = dict((k, dict((pk, np.zeros_like(pv)) for pk, pv in P[k].items())) for k in P)
    rs = np.random.RandomState(42)

    def net(k):
        return O.OracleAMLP(P[k], heads[k])

    def step(names, loss_fn, t, lr=1e-3):
        nets = dict((k, net(k)) for k in P)
        loss = loss_fn(nets)
        loss.backward()
        for k in names:
            P[k] = torch_adam(P[k], nets[k].grads(), M[k], V[k], t, lr)
        return float(loss)

    for it in range(steps):
        x, a, rew, ter = _batch(B, 50 + it)
        nx = _batch(B, 60 + it)[0]
        nx[ter > 0.5] = 0.0
        noise = dict(eps=torch.from_numpy(rs.randn(B, L).astype(np.float32)), z_target=torch.from_numpy(rs.randn(B * n, L).astype(np.float32)),
                     z_actor=torch.from_numpy(rs.randn(B, L).astype(np.float32)))
        m = bcq.update(*[torch.from_numpy(v).cuda() for v in (x, a, rew, nx, ter)], noise=noise)
        li = step(('imit_enc', 'imit_dec'), lambda N: O.cvae_loss(N['imit_enc'], N['imit_dec'], x, a, noise['eps'].numpy(), bcq.beta), it + 1)
        y = O.bcq_target(net('imit_dec'), net('policy_targ'), [net('q1_targ'), net('q2_targ')], nx, noise['z_target'].numpy(), n, bcq.scale,
                         bcq.lam, rew, ter, bcq.gamma)
        lc = step(('q1', 'q2'), lambda N: O.critic_loss([N['q1'], N['q2']], x, a, y), it + 1)
        la = step(('policy',), lambda N: O.actor_loss(N['imit_dec'], N['policy'], N['q1'], x, noise['z_actor'].numpy(), bcq.scale), it + 1)
        for k in ('policy', 'q1', 'q2'):
            P[k + '_targ'] = O.soft_sync(P[k + '_targ'], P[k], bcq.tau)
        assert abs(float(m['imitator_loss']) - li) < 2e-4 * max(1.0, abs(li)), (it, float(m['imitator_loss']), li)
        assert abs(float(m['critic_loss']) - lc) < 2e-3 * max(1.0, abs(lc)), (it, float(m['critic_loss']), lc)
        assert abs(float(m['actor_loss']) - la) < 2e-3 * max(1.0, abs(la)), (it, float(m['actor_loss']), la)
    for k in P:
        w = getattr(bcq, k).weights()
        for pk in P[k]:
            assert np.abs(w[pk].cpu().numpy() - P[k][pk]).max() < 2e-4, (k, pk, np.abs(w[pk].cpu().numpy() - P[k][pk]).max())
    assert bcq.total_step == steps
    bcq.close()


@pytest.mark.parametrize('nograd', ['fp32', 'fp16x2'])
def test_predict_is_the_best_sampled_action(nograd):
    import torch
    from oracle import offline_conti as O
    B, n = 100, 12                                # more observations than predict_rows: exercises the chunking
    bcq, orc = _learner_pair(51, 64, n, scale=dict(imit_dec=2.0, q1=2.0), nograd=nograd)       # spread the sampled actions and their values
    x = _batch(B, 52)[0]
    z = np.random.RandomState(53).randn(B * n, L).astype(np.float32)
    got = bcq.predict(torch.from_numpy(x).cuda(), noise=torch.from_numpy(z)).cpu().numpy()
    want, idx, v = O.predict_best_action(orc['imit_dec'], orc['policy'], orc['q1'], x, z, n, bcq.scale)
    srt = np.sort(v.numpy(), axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-4                     # rows whose best value is not a near-tie in fp32
    assert clear.sum() > B // 2, clear.sum()
    assert np.abs(got[clear] - want.numpy()[clear]).max() < 2e-4
    # every row: the device's pick is worth (in float64) what the best sampled action is worth, up to fp32 rounding
    got_v = orc['q1'](x, got)[:, 0].detach().numpy()
    assert np.abs(got_v - srt[:, -1]).max() < 2e-4
    assert (np.abs(got) <= 1.0).all()
    pv = bcq.predict_value(torch.from_numpy(x).cuda(), torch.from_numpy(got).cuda()).cpu().numpy()
    wv = 0.5 * (orc['q1'](x, got) + orc['q2'](x, got))[:, 0].detach().numpy()
    assert np.abs(pv - wv).max() < 2e-4 * max(1.0, np.abs(wv).max())
    bcq.close()


def _make_cfg(d, B=64, T=9):
    from rl4rs_amd import synth
    text = synth.make_catalog_text(seed=4)
    cpath = os.path.join(d, 'c.csv')
    synth.write_text(cpath, text)
    lpath = os.path.join(d, 'log.csv')
    synth.write_records(lpath, synth.make_records(300, pages=1, seed=2, hash_size=2000, special_ids=synth.special_ids_from_text(text)))
    return {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
            "category_feature_num": 21, "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9,
            "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": lpath,
            "iteminfo_file": cpath, "cache_size": 256, "model_seed": 3, "return_tensors": True,
            "support_d3rl_mask": True, "support_conti_env": True}


def test_fit_on_the_generated_continuous_dataset_then_knn_rollout(tmp_path):
    """configs[4] end to end on one GPU: the device env generates the continuous logged-policy dataset
    (data_generate_rl4rs_a_conti), BCQ fits it, and the learned policy's embeddings drive the continuous env, whose K-NN
    resolves them to legal items (batchrl_trainer.py:377-411 `evaluate`)."""
    import torch
    import rl4rs_amd
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from rl4rs_amd.offline import generate_offline_dataset
    from rl4rs_amd.offline_rl import BCQ, transitions_from_mdp
    cfg = _make_cfg(str(tmp_path))
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    data = generate_offline_dataset(env, epochs=6, shuffle=False)
    assert data['actions'].shape[1] == E and data['observations'].shape[1] == D
    tr = transitions_from_mdp(data['observations'], data['actions'], data['rewards'], data['terminals'], discrete_action=False)
    assert tr[1].dtype == torch.float32 and tr[1].shape[1] == E
    norms = tr[1].norm(dim=1)
    assert ((norms - 1).abs() < 1e-4).logical_or(norms == 0).all()        # logged actions are catalogue embeddings (or the pad row)
    bcq = BCQ(cfg, D, batch_size=256, n_action_samples=20, seed=1)
    hist = bcq.fit(tr, 60)
    for k in ('imitator_loss', 'critic_loss', 'actor_loss'):
        assert len(hist[k]) == 60 and np.isfinite(hist[k]).all(), k
    assert np.mean(hist['imitator_loss'][-10:]) < 0.6 * np.mean(hist['imitator_loss'][:5])      # the VAE learns the logged actions
    # policy -> K-NN rollout
    obs = env.reset()
    total = torch.zeros(cfg['batch_size'], dtype=torch.float64, device='cuda')
    for t in range(cfg['max_steps']):
        act = bcq.predict(obs)
        assert tuple(act.shape) == (cfg['batch_size'], E)
        obs, reward, done, info = env.step(act)
        total += torch.as_tensor(reward, device='cuda', dtype=torch.float64)
    prev = env.samples.prev_actions
    prev = prev.cpu().numpy() if torch.is_tensor(prev) else np.asarray(prev)
    loc = np.asarray(env.samples.location_mask)
    for j in range(9):
        assert (loc[j // 3][prev[:, j]] == 1).all()                       # the K-NN only resolves to items legal for the slot
    assert all(len(set(r)) == 9 for r in prev.tolist())                   # and never repeats an item within the slate
    assert torch.isfinite(total).all() and bool(done[0] if not isinstance(done, (int, bool)) else done)
    bcq.close()


def _bcq_worker(rank, world, port, d, out):
    import torch
    from test_gpu_train_dp import _init_dist
    Dm = _init_dist(rank, world, port)
    from rl4rs_amd.offline_rl import BCQ
    data = torch.load(os.path.join(d, 'bcq_data.pt'), weights_only=False)
    bcq = BCQ({'action_emb_size': E}, D, batch_size=32, n_action_samples=6, seed=4)
    for k in range(2):
        b = data[rank][k]
        bcq.update(*[b[n].cuda() for n in ('obs', 'act', 'rew', 'nxt', 'ter')], noise=dict((n, b[n]) for n in ('eps', 'z_target', 'z_actor')))
    w = dict((name, dict((k, v.cpu()) for k, v in getattr(bcq, name).weights().items())) for name in ('imit_enc', 'imit_dec', 'policy', 'q1', 'q2', 'q1_targ'))
    Dm.barrier()
    torch.save(w, os.path.join(d, 'bcq%d.pt' % rank))
    out.put(rank)


def test_bcq_two_ranks_on_one_gpu(tmp_path):
    """data-parallel BCQ at world_size 2 (gloo, both ranks on the one GPU): replicas bit-identical after two updates, and equal to
    ONE process training on the concatenated minibatches (every loss is a batch mean and the target is per row, so the mean
    of two half-batch gradients is the full-batch gradient)."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_train_dp import _spawn
    from rl4rs_amd.offline_rl import BCQ
    d = str(tmp_path)
    B, n = 32, 6
    rs = np.random.RandomState(0)

    def mb(seed):
        x, a, rew, ter = _batch(B, seed)
        nx = _batch(B, seed + 100)[0]
        f = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32))
        return dict(obs=f(x), act=f(a), rew=f(rew), nxt=f(nx), ter=f(ter), eps=f(rs.randn(B, L)), z_target=f(rs.randn(B * n, L)),
                    z_actor=f(rs.randn(B, L)))

    data = [[mb(1), mb(2)], [mb(3), mb(4)]]
    torch.save(data, os.path.join(d, 'bcq_data.pt'))
    _spawn(_bcq_worker, (d,))
    w0 = torch.load(os.path.join(d, 'bcq0.pt'), weights_only=False)
    w1 = torch.load(os.path.join(d, 'bcq1.pt'), weights_only=False)
    for name in w0:
        for k in w0[name]:
            assert torch.equal(w0[name][k], w1[name][k]), (name, k)
    one = BCQ({'action_emb_size': E}, D, batch_size=2 * B, n_action_samples=n, seed=4)
    for k in range(2):
        a, b = data[0][k], data[1][k]
        cat = dict((key, torch.cat([a[key], b[key]]).contiguous()) for key in a)
        one.update(*[cat[key].cuda() for key in ('obs', 'act', 'rew', 'nxt', 'ter')], noise=dict((key, cat[key]) for key in ('eps', 'z_target', 'z_actor')))
    for name in w0:
        ref = getattr(one, name).weights()
        for k in w0[name]:
            assert (w0[name][k] - ref[k].cpu()).abs().max().item() < 5e-5, (name, k)
    one.close()


@pytest.mark.parametrize('nograd', ['fp32', 'fp16x2'])
def test_bcq_at_the_bench_shape(tmp_path, nograd):
    """The shapes bench.py and the reference run (script/batchrl_trainer.py:61-73 batch_size 256, d3rlpy's 100 action samples;
    :377-411 evaluate): ONE update of 256 transitions x 100 sampled actions (25 600 target rows through four networks) against the
    float64 restatement, bars as in test_updates_track_the_fp64_restatement; then predict over 4 096 observations x 100 (409 600
    rows, ONE pass: predict_rows = the env batch like bench.py's BcqWorkload) - the restatement's pick on a 256-observation subset,
    and on ALL rows the size-independent properties: |a| <= 1, the pick's float64 value is the row maximum of the float64 values
    of the 100 sampled actions to 2e-4, and the masked K-NN resolves every pick to a legal item (== the numpy rule)."""
    import torch
    from oracle import offline_conti as O
    from oracle.offline_rl import torch_adam
    from oracle.state import nearest_neighbor_with_mask
    from rl4rs_amd import device as Dv
    from rl4rs_amd import synth
    from rl4rs_amd.data import CatalogTables
    from rl4rs_amd.offline_rl import BCQ
    B, n, NP = 256, 100, 4096
    bcq = BCQ({'action_emb_size': E}, D, batch_size=B, n_action_samples=n, predict_rows=NP, seed=43, nograd_precision=nograd)
    if nograd == 'fp16x2':
        assert all(net.h16_ok for net in bcq.nets) and B * n >= bcq.q1.H16_MIN_ROWS        # the fused kernel takes these rows by itself
    # spread the sampled actions and their values (fresh networks value every action almost alike)
    for k, f in dict(imit_dec=2.0, q1=2.0, q1_targ=2.0, q2_targ=1.5).items():
        net = getattr(bcq, k)
        net.set_flat_params((net.flat_params() * f).contiguous())
    names = ('imit_enc', 'imit_dec', 'policy', 'policy_targ', 'q1', 'q2', 'q1_targ', 'q2_targ')
    heads = dict(imit_dec='tanh', policy='tanh', policy_targ='tanh')
    P = dict((k, dict((pk, pv.cpu().numpy().astype(np.float64)) for pk, pv in getattr(bcq, k).weights().items())) for k in names)
    net = lambda k: O.OracleAMLP(P[k], heads.get(k, 'none'))
    # ---- one update
    rs = np.random.RandomState(44)
    x, a, rew, ter = _batch(B, 70)
    nx = _batch(B, 71)[0]
    nx[ter > 0.5] = 0.0
    noise = dict(eps=torch.from_numpy(rs.randn(B, L).astype(np.float32)), z_target=torch.from_numpy(rs.randn(B * n, L).astype(np.float32)),
                 z_actor=torch.from_numpy(rs.randn(B, L).astype(np.float32)))
    m = bcq.update(*[torch.from_numpy(v).cuda() for v in (x, a, rew, nx, ter)], noise=noise)
    M = dict((k, dict((pk, np.zeros_like(pv)) for pk, pv in P[k].items())) for k in P)
    V = dict((k, dict((pk, np.zeros_like(pv)) for pk, pv in P[k].items())) for k in P)

    GR = {}

    def step(step_names, loss_fn):
        nets = dict((k, net(k)) for k in P)
        loss = loss_fn(nets)
        loss.backward()
        for k in step_names:
            GR[k] = nets[k].grads()
            P[k] = torch_adam(P[k], GR[k], M[k], V[k], 1, 1e-3)
        return float(loss.detach())

    li = step(('imit_enc', 'imit_dec'), lambda N: O.cvae_loss(N['imit_enc'], N['imit_dec'], x, a, noise['eps'].numpy(), bcq.beta))
    y = O.bcq_target(net('imit_dec'), net('policy_targ'), [net('q1_targ'), net('q2_targ')], nx, noise['z_target'].numpy(), n, bcq.scale,
                     bcq.lam, rew, ter, bcq.gamma)
    lc = step(('q1', 'q2'), lambda N: O.critic_loss([N['q1'], N['q2']], x, a, y))
    la = step(('policy',), lambda N: O.actor_loss(N['imit_dec'], N['policy'], N['q1'], x, noise['z_actor'].numpy(), bcq.scale))
    for k in ('policy', 'q1', 'q2'):
        P[k + '_targ'] = O.soft_sync(P[k + '_targ'], P[k], bcq.tau)
    assert abs(float(m['imitator_loss']) - li) < 2e-4 * max(1.0, abs(li)), (float(m['imitator_loss']), li)
    assert abs(float(m['critic_loss']) - lc) < 2e-3 * max(1.0, abs(lc)), (float(m['critic_loss']), lc)
    assert abs(float(m['actor_loss']) - la) < 2e-3 * max(1.0, abs(la)), (float(m['actor_loss']), la)
    # the FIRST Adam step is lr * g / (|g| + 1e-8): wherever |g| is far above 1e-8 it is +-lr whatever the gradient's last bits are,
    # but an element whose float64 gradient is itself at the 1e-8 scale moves by a fraction of lr that depends on those bits.
    # Such elements (saturated tanh heads, the actor's first layer) are held to what two steps of opposite sign can differ by (2 lr), the rest - most of every network - to 2e-4.
    n_tight = n_all = 0
    for k in P:
        w = getattr(bcq, k).weights()
        for pk in P[k]:
            err = np.abs(w[pk].cpu().numpy() - P[k][pk])
            tiny = ((np.abs(GR[k][pk]) < 2e-6) & (GR[k][pk] != 0)) if k in GR else np.zeros(err.shape, bool)       # (exact zeros: dead ReLU units, no step on either side)
            n_tight += int((~tiny).sum())
            n_all += tiny.size
            assert err[~tiny].max() < 2e-4, (k, pk, err[~tiny].max())
            assert err.max() <= 2.001e-3, (k, pk, err.max())                 # (a gradient that rounds to the other sign: +lr against -lr)
    assert n_tight > 0.8 * n_all, (n_tight, n_all)
    # ---- predict over the whole env batch: the restatement takes the device's updated parameters (the 2e-4 parameter bar above
    # is wider than the 2e-4 action bar below)
    P = dict((k, dict((pk, pv.cpu().numpy().astype(np.float64)) for pk, pv in getattr(bcq, k).weights().items())) for k in names)
    xs = _batch(NP, 72)[0]
    z = np.random.RandomState(73).randn(NP * n, L).astype(np.float32)
    got_t = bcq.predict(torch.from_numpy(xs).cuda(), noise=torch.from_numpy(z))
    got = got_t.cpu().numpy()
    assert got.shape == (NP, E) and np.isfinite(got).all() and (np.abs(got) <= 1.0).all()
    dec, pol, q1 = net('imit_dec'), net('policy'), net('q1')
    S = 256
    want, idx, v = O.predict_best_action(dec, pol, q1, xs[:S], z[:S * n], n, bcq.scale)
    srt = np.sort(v.numpy(), axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-4                   # rows whose best value is not a near-tie in fp32
    assert clear.sum() > S // 2, clear.sum()
    assert np.abs(got[:S][clear] - want.numpy()[clear]).max() < 2e-4
    # every row: the pick is worth (in float64) what the best of its 100 sampled actions is worth
    worst = 0.0
    for lo in range(0, NP, 512):
        _, _, vv = O.predict_best_action(dec, pol, q1, xs[lo:lo + 512], z[lo * n:(lo + 512) * n], n, bcq.scale)
        with torch.no_grad():
            gv = q1(xs[lo:lo + 512], got[lo:lo + 512])[:, 0].numpy()
        worst = max(worst, float(np.abs(gv - vv.numpy().max(axis=1)).max()))
    assert worst < 2e-4, worst
    # the masked K-NN of the continuous env resolves every pick to a legal item - and to the item numpy's rule picks
    cpath = os.path.join(str(tmp_path), 'c.csv')
    synth.write_text(cpath, synth.make_catalog_text(seed=4))
    tab = CatalogTables(cpath, 284, E)
    mask = (np.random.RandomState(74).rand(NP, 284) < 0.3).astype(np.int64) * np.asarray(tab.location_mask)[np.arange(NP) % 3]
    r = np.arange(NP)
    mask[r, 1 + r % 39] |= (r % 3 == 0)                       # never an empty row
    mask[r, 40 + r % 108] |= (r % 3 == 1)
    mask[r, 148 + r % 136] |= (r % 3 == 2)
    item = Dv.knn(got_t, torch.from_numpy(tab.action_emb).cuda(), mask=mask).cpu().numpy()
    assert (mask[np.arange(NP), item] == 1).all()
    ref_item = nearest_neighbor_with_mask(got.astype(np.float64), tab.action_emb, mask)
    # (a float32 action is exact in float64; the catalogue rows are float64 on both sides: the integer pick is the same)
    assert np.array_equal(item, ref_item)
    bcq.close()
