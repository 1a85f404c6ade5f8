"""The continuous CQL learner ('CQL-conti', script/batchrl_trainer.py:91-107 = d3rlpy.algos.CQL on default encoders) against the
float64 torch restatement in oracle/offline_conti.py (PARITY UNPINNED: d3rlpy is absent): squashed-Gaussian sampling and its
log-probability, the conservative term and every critic gradient, the actor gradient through min(Q1, Q2), the temperature and
alpha steps, whole updates with supplied noise tracked for several steps, an end-to-end fit on the generated continuous dataset.

Tolerances as in test_gpu_offline_conti.py: values 2e-4 abs on O(1) quantities, gradients 2e-3 of each array's largest entry."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_offline_conti import D, E, _batch, _close, _make_cfg          # noqa: E402

pytestmark = pytest.mark.gpu


def _learner_pair(seed, B, n, **kw):
    from oracle.offline_conti import OracleAMLP
    from rl4rs_amd.offline_rl import CQL
    cql = CQL({'action_emb_size': E}, D, batch_size=B, n_action_samples=n, seed=seed, **kw)
    # a policy with some spread (default-initialised heads give sigma ~ 1, mu ~ 0.05) and critics that disagree
    for name, f in (('policy', 1.5), ('q2', 1.2), ('q1_targ', 0.8), ('q2_targ', 1.1)):
        net = getattr(cql, name)
        net.set_flat_params((net.flat_params() * f).contiguous())
    orc = dict((k, OracleAMLP(dict((pk, pv.cpu().numpy()) for pk, pv in getattr(cql, k).weights().items())))
               for k in ('policy', 'q1', 'q2', 'q1_targ', 'q2_targ'))
    return cql, orc


def _noise(rs, B, n):
    import torch
    f = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32))
    u = lambda: torch.from_numpy(rs.uniform(-1, 1, size=(B, n, E)).astype(np.float32))
    return dict(eps_temp=f(B, E), alpha=(f(B * n, E), f(B * n, E), u()), critic=(f(B * n, E), f(B * n, E), u()), eps_actor=f(B, E))


def test_squashed_sample_and_log_prob():
    import torch
    from oracle import offline_conti as O
    from rl4rs_amd import device as Dv
    cql, orc = _learner_pair(3, 32, 4)
    x = _batch(32, 5)[0]
    rs = np.random.RandomState(6)
    for rep in (1, 4):
        eps = (rs.randn(32 * rep, E) * 1.5).astype(np.float32)
        head = cql.policy.forward(torch.from_numpy(x).cuda())
        a, lp = Dv.squashed_sample(head, torch.from_numpy(eps).cuda(), rep=rep)
        wa, wlp = O.squashed_sample(orc['policy'], x, eps)
        assert np.abs(a.cpu().numpy() - wa.detach().numpy()).max() < 5e-5           # tanh(mu + exp(logstd) * 1.5 randn): the fp32 head's ~1e-5 x the noise scale
        assert np.abs(lp.cpu().numpy() - wlp.detach().numpy()).max() < 2e-3          # sums of 32 terms of size O(1..10)
    # destination layout: sample groups side by side, untouched columns stay as they were
    acts = torch.full((32, 7, E), 9.0, device='cuda')
    lps = torch.full((32, 7), 9.0, device='cuda')
    eps = rs.randn(32 * 2, E).astype(np.float32)
    Dv.squashed_sample(head, torch.from_numpy(eps).cuda(), rep=2, act_out=acts.view(-1, E), logp_out=lps.view(-1), out_rep=7, out_off=3)
    wa, wlp = O.squashed_sample(orc['policy'], x, eps)
    assert np.abs(acts[:, 3:5].reshape(-1, E).cpu().numpy() - wa.detach().numpy()).max() < 5e-5
    assert (acts[:, :3] == 9).all() and (acts[:, 5:] == 9).all() and (lps[:, :3] == 9).all() and (lps[:, 5:] == 9).all()
    # deterministic head
    a, lp = Dv.squashed_sample(head, None)
    assert lp is None and np.abs(a.cpu().numpy() - O.best_action(orc['policy'], x).detach().numpy()).max() < 2e-5
    assert np.abs(cql.predict(torch.from_numpy(x).cuda()).cpu().numpy() - O.best_action(orc['policy'], x).detach().numpy()).max() < 2e-5
    cql.close()


def test_log_prob_is_stable_at_saturated_actions():
    """softplus(-2u) in its overflow-safe form: |u| up to 40 gives a finite log-prob equal to the float64 value"""
    import torch
    from rl4rs_amd import device as Dv
    head = torch.zeros(4, 2 * E, device='cuda')
    head[:, :E] = torch.tensor([-40.0, -3.0, 3.0, 40.0], device='cuda')[:, None]
    head[:, E:] = -3.0
    eps = torch.zeros(4, E, device='cuda')
    a, lp = Dv.squashed_sample(head, eps)
    u = head[:, :E].double().cpu()
    want = (-(-3.0) - 0.5 * np.log(2 * np.pi) - 2 * (np.log(2.0) - u - torch.nn.functional.softplus(-2 * u))).sum(dim=1)
    assert torch.isfinite(lp).all() and np.allclose(lp.cpu().numpy(), want.numpy(), rtol=1e-5)
    assert (a.abs() <= 1).all()


def test_critic_actor_temperature_and_alpha_gradients():
    import torch
    from oracle import offline_conti as O
    from rl4rs_amd import device as Dv
    B, n = 48, 5
    cql, orc = _learner_pair(11, B, n, gamma=1.0)
    m = cql.m
    x, a, rew, ter = _batch(B, 12)
    nx = _batch(B, 13)[0]
    nx[ter > 0.5] = 0.0
    nz = _noise(np.random.RandomState(14), B, n)
    xd, ad, nd, rd, td = [torch.from_numpy(v).cuda() for v in (x, a, nx, rew, ter)]
    head_nxt = cql.policy.forward(nd)
    head_obs = cql.policy.forward(xd)
    log_alpha = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    cql.log_alpha.p.fill_(0.3)
    log_temp = torch.tensor([-0.2], dtype=torch.float64, requires_grad=True)
    cql.log_temp.p.fill_(-0.2)
    # target
    a_next, _ = Dv.squashed_sample(head_nxt, None)
    y, _ = Dv.bcq_target(cql.q1_targ.forward(nd, a_next), cql.q2_targ.forward(nd, a_next), 1, 1.0, rd, td, cql.gamma)
    want_y = O.cql_target(orc['policy'], [orc['q1_targ'], orc['q2_targ']], nx, rew, ter, cql.gamma)
    assert np.abs(y.cpu().numpy() - want_y.numpy()).max() < 5e-4 * max(1.0, float(want_y.abs().max()))
    # critic: TD + conservative term, both critics
    fa, fo = cql._conservative_rows(head_obs, head_nxt, ad, nz['critic'])
    aw = (cql.log_alpha.p.exp().clamp(0, 1e6) * cql.conservative_weight).contiguous()
    sums, dq1, dq2 = Dv.cql_critic_loss(cql.q1.forward(xd, fa, rep=m), cql.q2.forward(xd, fa, rep=m), fo, m, y=y, alpha_w=aw)
    cql.q1.backward(xd, fa, dq1, rep=m)
    cql.q2.backward(xd, fa, dq2, rep=m)
    e_t, e_tp1, uni = [v.numpy() for v in nz['critic']]
    cons = O.conservative_loss(orc['policy'], [orc['q1'], orc['q2']], log_alpha, x, a, nx, e_t, e_tp1, uni, n, cql.conservative_weight,
                               cql.alpha_threshold)
    want = O.critic_loss([orc['q1'], orc['q2']], x, a, want_y) + cons
    want.backward()
    got = float((sums[0] + sums[1]) / B + np.exp(0.3) * (cql._conservative_value(sums, B) - cql.alpha_threshold))
    assert abs(got - float(want.detach())) < 1e-3 * max(1.0, abs(float(want.detach()))), (got, float(want.detach()))
    for k in ('q1', 'q2'):
        g, gw = getattr(cql, k).gradients(), orc[k].grads()
        for pk in gw:
            _close(g[pk].cpu().numpy(), gw[pk], 2e-3, 'critic %s grad %s' % (k, pk))
    # alpha: d(-conservative)/d log_alpha
    want_ga = -float(log_alpha.grad)
    got_ga = float(-(np.exp(0.3) * (cql._conservative_value(sums, B) - cql.alpha_threshold)))
    assert abs(got_ga - want_ga) < 1e-3 * max(1.0, abs(want_ga))
    # actor through min(Q1, Q2)
    for v in orc.values():
        v.zero_grad()
    eps = nz['eps_actor'].cuda()
    a_pi, logp = Dv.squashed_sample(head_obs, eps)
    qmin, d1, d2 = Dv.twin_min(cql.q1.forward(xd, a_pi), cql.q2.forward(xd, a_pi), want_grad=True)
    g_a = cql.q1.backward(xd, a_pi, d1.view(B, 1), want_dact=True, want_param_grad=False)
    g_a += cql.q2.backward(xd, a_pi, d2.view(B, 1), want_dact=True, want_param_grad=False)
    cql.policy.backward(xd, None, Dv.sac_actor_grad(head_obs, eps, a_pi, g_a, cql.log_temp.p))
    want = O.sac_actor_loss(orc['policy'], [orc['q1'], orc['q2']], log_temp, x, nz['eps_actor'].numpy())
    want.backward()
    got = float((cql.log_temp.p.exp() * logp - qmin).mean())
    assert abs(got - float(want.detach())) < 2e-4 * max(1.0, abs(float(want.detach())))
    g, gw = cql.policy.gradients(), orc['policy'].grads()
    for pk in gw:
        _close(g[pk].cpu().numpy(), gw[pk], 2e-3, 'actor grad %s' % pk)
    sel = (d1 != 0).sum().item()
    assert 0 < sel < B                                      # both critics are the minimum somewhere: the selector is exercised
    # temperature
    want = O.temp_loss(orc['policy'], log_temp, x, nz['eps_temp'].numpy())
    want.backward()
    _, lp = Dv.squashed_sample(head_obs, nz['eps_temp'].cuda())
    got_g = float(-(np.exp(-0.2) * (lp - E).mean()))
    assert abs(got_g - float(log_temp.grad)) < 1e-3 * max(1.0, abs(float(log_temp.grad)))
    cql.close()


def test_updates_track_the_fp64_restatement():
    """three whole updates (temperature, alpha, critic, actor, soft target update) with the same noise on both sides"""
    import torch
    from oracle import offline_conti as O
    from oracle.offline_rl import torch_adam
    B, n, steps = 48, 4, 3
    cql, orc = _learner_pair(21, B, n, gamma=1.0)
    P = dict((k, v.numpy_params()) for k, v in orc.items())
    M = dict((k, dict((pk, np.zeros_like(pv)) for pk, pv in P[k].items())) for k in P)
    V = dict((k, dict((pk, np.zeros_like(pv)) for pk, pv in P[k].items())) for k in P)
    S = dict(log_temp=np.array([0.0]), log_alpha=np.array([0.0]))
    SM = dict((k, np.zeros(1)) for k in S)
    SV = dict((k, np.zeros(1)) for k in S)
    rs = np.random.RandomState(22)
    lrs = dict(policy=cql.actor_lr, q1=cql.critic_lr, q2=cql.critic_lr)

    def net(k):
        return O.OracleAMLP(P[k])

    for it in range(steps):
        x, a, rew, ter = _batch(B, 70 + it)
        nx = _batch(B, 80 + it)[0]
        nx[ter > 0.5] = 0.0
        nz = _noise(rs, B, n)
        mt = cql.update(*[torch.from_numpy(v).cuda() for v in (x, a, rew, nx, ter)], noise=nz)
        t = it + 1
        # temperature
        lt = torch.tensor(S['log_temp'], requires_grad=True)
        l_temp = O.temp_loss(net('policy'), lt, x, nz['eps_temp'].numpy())
        l_temp.backward()
        SMd, SVd = {'x': SM['log_temp']}, {'x': SV['log_temp']}
        S['log_temp'] = torch_adam({'x': S['log_temp']}, {'x': lt.grad.numpy()}, SMd, SVd, t, cql.temp_lr)['x']
        SM['log_temp'], SV['log_temp'] = SMd['x'], SVd['x']
        # alpha
        la = torch.tensor(S['log_alpha'], requires_grad=True)
        e_t, e_tp1, uni = [v.numpy() for v in nz['alpha']]
        l_alpha = -O.conservative_loss(net('policy'), [net('q1'), net('q2')], la, x, a, nx, e_t, e_tp1, uni, n, cql.conservative_weight,
                                       cql.alpha_threshold)
        l_alpha.backward()
        SMd, SVd = {'x': SM['log_alpha']}, {'x': SV['log_alpha']}
        S['log_alpha'] = torch_adam({'x': S['log_alpha']}, {'x': la.grad.numpy()}, SMd, SVd, t, cql.alpha_lr)['x']
        SM['log_alpha'], SV['log_alpha'] = SMd['x'], SVd['x']
        # critic
        y = O.cql_target(net('policy'), [net('q1_targ'), net('q2_targ')], nx, rew, ter, cql.gamma)
        nets = dict((k, net(k)) for k in P)
        e_t, e_tp1, uni = [v.numpy() for v in nz['critic']]
        l_c = O.critic_loss([nets['q1'], nets['q2']], x, a, y) + O.conservative_loss(
            nets['policy'], [nets['q1'], nets['q2']], torch.tensor(S['log_alpha']), x, a, nx, e_t, e_tp1, uni, n, cql.conservative_weight,
            cql.alpha_threshold)
        l_c.backward()
        for k in ('q1', 'q2'):
            P[k] = torch_adam(P[k], nets[k].grads(), M[k], V[k], t, lrs[k])
        # actor
        nets = dict((k, net(k)) for k in P)
        l_a = O.sac_actor_loss(nets['policy'], [nets['q1'], nets['q2']], torch.tensor(S['log_temp']), x, nz['eps_actor'].numpy())
        l_a.backward()
        P['policy'] = torch_adam(P['policy'], nets['policy'].grads(), M['policy'], V['policy'], t, lrs['policy'])
        for k in ('q1', 'q2'):
            P[k + '_targ'] = O.soft_sync(P[k + '_targ'], P[k], cql.tau)
        for name, ref in (('temp_loss', l_temp), ('alpha_loss', l_alpha), ('critic_loss', l_c), ('actor_loss', l_a)):
            ref = float(ref.detach())
            assert abs(float(mt[name]) - ref) < 2e-3 * max(1.0, abs(ref)), (it, name, float(mt[name]), ref)
    for k in P:
        w = getattr(cql, k).weights()
        for pk in P[k]:
            assert np.abs(w[pk].cpu().numpy() - P[k][pk]).max() < 2e-4, (k, pk, np.abs(w[pk].cpu().numpy() - P[k][pk]).max())
    assert abs(float(cql.log_temp.p) - float(S['log_temp'][0])) < 1e-5 and abs(float(cql.log_alpha.p) - float(S['log_alpha'][0])) < 1e-5
    assert float(cql.log_temp.p) != 0.0 and float(cql.log_alpha.p) != 0.0
    cql.close()


def test_update_as_one_library_call_equals_the_per_phase_calls():
    """rl4rs_cql_update (the whole update as one host call with the learned scalars' Adam on the device; the default on one rank)
    against CQL.update's per-phase path over four updates with shared noise: the networks bit-identical wherever the two paths run
    the same kernels, the scalars and everything downstream of them equal to float32 rounding (device expf against torch.exp)"""
    import torch
    B, n = 32, 4
    a, _ = _learner_pair(71, B, n, gamma=1.0)
    b, _ = _learner_pair(71, B, n, gamma=1.0)
    assert a.one_call
    b.one_call = False
    rs = np.random.RandomState(72)
    f = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32)).cuda()
    for it in range(4):
        x, act, rew, ter = _batch(B, 90 + it)
        nx = _batch(B, 95 + it)[0]
        noise = _noise(rs, B, n)
        args = [f(v) for v in (x, act, rew, nx, ter)]
        ma = a.update(*args, noise=noise)
        mb = b.update(*args, noise=noise)
        assert set(ma) == set(mb) == {'critic_loss', 'actor_loss', 'temp_loss', 'alpha_loss'}
        for k in ma:
            assert abs(float(ma[k]) - float(mb[k])) <= 1e-5 * max(1.0, abs(float(mb[k]))), (it, k, float(ma[k]), float(mb[k]))
    for sa, sb in ((a.log_temp, b.log_temp), (a.log_alpha, b.log_alpha)):
        assert sa.t == sb.t == 4
        assert torch.allclose(sa.state[:3], sb.state[:3], rtol=1e-4, atol=1e-6), (sa.state, sb.state)      # (a value that has walked +-lr around its start: absolute bar)
    for na, nb in zip(a.nets, b.nets):
        assert (na.flat_params() - nb.flat_params()).abs().max().item() < 1e-5
    assert a.total_step == b.total_step == 4
    a.close()
    b.close()


def test_one_call_update_validates_inputs_and_takes_partial_noise():
    """the one-call update asserts device / dtype / shape like the per-phase path, and a noise dict with only some of
    the keys is completed from the generator (the per-phase path reads it with noise.get) instead of raising KeyError"""
    import torch
    B, n = 32, 4
    a, _ = _learner_pair(73, B, n, gamma=1.0)
    b, _ = _learner_pair(73, B, n, gamma=1.0)
    rs = np.random.RandomState(74)
    f = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32)).cuda()
    x, act, rew, ter = _batch(B, 91)
    nx = _batch(B, 96)[0]
    good = [f(v) for v in (x, act, rew, nx, ter)]
    for args in ([good[0].cpu()] + good[1:], good[:2] + [good[2].double()] + good[3:], good[:4] + [good[4] > 0.5],
                 [good[0][:, :-1]] + good[1:], good[:1] + [good[1][:, :-1]] + good[2:]):
        with pytest.raises(AssertionError):
            a.update(*args)
    assert a.total_step == 0
    full = _noise(rs, B, n)
    # all keys given == the same dict through the old all-or-nothing branch (b): bit-identical networks
    a.update(*good, noise=dict(full))
    b.update(*good, noise=full)
    for na, nb in zip(a.nets, b.nets):
        assert torch.equal(na.flat_params(), nb.flat_params())
    # a partial dict: runs, and draws the rest from the learner's generator (the step differs from the fully specified one)
    part = {'eps_temp': full['eps_temp'], 'critic': full['critic']}
    m = a.update(*good, noise=part)
    assert set(m) == {'critic_loss', 'actor_loss', 'temp_loss', 'alpha_loss'} and all(np.isfinite(float(v)) for v in m.values())
    with pytest.raises(AssertionError):
        a.update(*good, noise={'eps_temp': full['eps_temp'][:-1]})
    a.close()
    b.close()


def test_fit_on_the_generated_continuous_dataset_then_knn_rollout(tmp_path):
    """'CQL-conti' end to end on one GPU as the script configures it (gamma = 1, standard reward scaler): fit the continuous
    logged-policy dataset the device env generates, then drive the env with tanh(mu(s)) through the K-NN."""
    import torch
    import rl4rs_amd
    from rl4rs.policy.policy_model import policy_model
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    from rl4rs_amd.offline import generate_offline_dataset
    from rl4rs_amd.offline_rl import CQL, StandardRewardScaler, transitions_from_mdp
    cfg = _make_cfg(str(tmp_path))
    env = rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))
    data = generate_offline_dataset(env, epochs=6, shuffle=False)
    tr = transitions_from_mdp(data['observations'], data['actions'], data['rewards'], data['terminals'], discrete_action=False)
    scaler = StandardRewardScaler(data['rewards'])
    z = scaler.transform(data['rewards'])
    assert abs(float(z.mean())) < 1e-3 and abs(float(z.std(unbiased=False)) - 1.0) < 1e-2
    cql = CQL(cfg, D, batch_size=256, gamma=1.0, reward_scaler=scaler, seed=1)
    t0, a0 = float(cql.log_temp.p), float(cql.log_alpha.p)
    hist = cql.fit(tr, 40)
    for k in ('critic_loss', 'actor_loss', 'temp_loss', 'alpha_loss'):
        assert len(hist[k]) == 40 and np.isfinite(hist[k]).all(), k
    assert float(cql.log_temp.p) != t0 and float(cql.log_alpha.p) != a0
    policy = policy_model(cql, config=cfg)
    obs = env.reset()
    for t in range(cfg['max_steps']):
        act = policy.predict_with_mask(obs)
        assert tuple(act.shape) == (cfg['batch_size'], E) and bool((act.abs() <= 1).all())
        obs, reward, done, info = env.step(act)
    prev = env.samples.prev_actions
    prev = prev.cpu().numpy() if torch.is_tensor(prev) else np.asarray(prev)
    loc = np.asarray(env.samples.location_mask)
    for j in range(9):
        assert (loc[j // 3][prev[:, j]] == 1).all()
    assert all(len(set(r)) == 9 for r in prev.tolist())
    cql.close()


def _cql_worker(rank, world, port, d, out):
    import torch
    from test_gpu_train_dp import _init_dist
    Dm = _init_dist(rank, world, port)
    from rl4rs_amd.offline_rl import CQL
    data = torch.load(os.path.join(d, 'cql_data.pt'), weights_only=False)
    cql = CQL({'action_emb_size': E}, D, batch_size=24, n_action_samples=3, gamma=1.0, seed=4)
    for k in range(2):
        b = data[rank][k]
        cql.update(*[b[n].cuda() for n in ('obs', 'act', 'rew', 'nxt', 'ter')], noise=b['noise'])
    w = dict((name, dict((k, v.cpu()) for k, v in getattr(cql, name).weights().items())) for name in ('policy', 'q1', 'q2', 'q1_targ'))
    w['scalars'] = dict(log_temp=cql.log_temp.p.cpu(), log_alpha=cql.log_alpha.p.cpu())
    Dm.barrier()
    torch.save(w, os.path.join(d, 'cql%d.pt' % rank))
    out.put(rank)


def test_cql_two_ranks_on_one_gpu(tmp_path):
    """data-parallel CQL at world_size 2 (gloo, both ranks on the one GPU): replicas (networks and the two learned scalars)
    bit-identical after two updates, and equal to ONE process training on the concatenated minibatches."""
    import torch
    from test_gpu_train_dp import _spawn
    from rl4rs_amd.offline_rl import CQL
    d = str(tmp_path)
    B, n = 24, 3
    rs = np.random.RandomState(0)

    def mb(seed):
        x, a, rew, ter = _batch(B, seed)
        nx = _batch(B, seed + 100)[0]
        f = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32))
        return dict(obs=f(x), act=f(a), rew=f(rew), nxt=f(nx), ter=f(ter), noise=_noise(rs, B, n))

    data = [[mb(1), mb(2)], [mb(3), mb(4)]]
    torch.save(data, os.path.join(d, 'cql_data.pt'))
    _spawn(_cql_worker, (d,))
    w0 = torch.load(os.path.join(d, 'cql0.pt'), weights_only=False)
    w1 = torch.load(os.path.join(d, 'cql1.pt'), weights_only=False)
    for name in w0:
        for k in w0[name]:
            assert torch.equal(w0[name][k], w1[name][k]), (name, k)
    one = CQL({'action_emb_size': E}, D, batch_size=2 * B, n_action_samples=n, gamma=1.0, seed=4)
    for k in range(2):
        a, b = data[0][k], data[1][k]
        cat = dict((key, torch.cat([a[key], b[key]]).contiguous()) for key in ('obs', 'act', 'rew', 'nxt', 'ter'))
        nz = {}
        for key in ('eps_temp', 'eps_actor'):
            nz[key] = torch.cat([a['noise'][key], b['noise'][key]])
        for key in ('alpha', 'critic'):
            nz[key] = tuple(torch.cat([a['noise'][key][i], b['noise'][key][i]]) for i in range(3))
        one.update(*[cat[key].cuda() for key in ('obs', 'act', 'rew', 'nxt', 'ter')], noise=nz)
    for name in ('policy', 'q1', 'q2', 'q1_targ'):
        ref = getattr(one, name).weights()
        for k in w0[name]:
            assert (w0[name][k] - ref[k].cpu()).abs().max().item() < 5e-5, (name, k)
    assert abs(float(w0['scalars']['log_temp']) - float(one.log_temp.p)) < 1e-6
    assert abs(float(w0['scalars']['log_alpha']) - float(one.log_alpha.p)) < 1e-6
    one.close()
