"""GPU: the data-parallel training paths, run for real.

Two ranks on the ONE GPU of the box over gloo (gloo all-reduces through the host; RCCL refuses two ranks on one device):
``Trainer`` A2C and PPO, ``RawStateTrainer`` and an offline learner on its real device network.  Required: parameters
bit-identical across the ranks after every train call, and equal to a single-process run whose gradient is the mean of the
two ranks' shard gradients.  Plus PPO fidelity: ``Trainer`` tracks a float64 restatement of RLlib's train call (minibatch
SGD + Adam + adaptive kl_coeff) for three iterations.  Reference: script/modelfree_train.py:179-304,409."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_cfg(d, rank, B=64, T=9, seq=False, raw=False):
    from rl4rs_amd import synth
    text = synth.make_catalog_text(seed=4)
    cpath = os.path.join(d, 'c.csv')
    if not os.path.exists(cpath):
        synth.write_text(cpath, text)
    lpath = os.path.join(d, 'log%d.csv' % rank)
    recs = synth.make_records(300, pages=4 if seq else 1, seed=2 + rank, hash_size=2000, special_ids=synth.special_ids_from_text(text))
    synth.write_records(lpath, recs)
    cfg = {"maxlen": 64, "batch_size": B, "action_size": 284, "class_num": 2, "dense_feature_num": 432,
           "category_feature_num": 21, "category_hash_size": 2000, "seq_num": 2, "emb_size": 128, "page_items": 9,
           "hidden_units": 128, "max_steps": T, "action_emb_size": 32, "sample_file": lpath,
           "iteminfo_file": cpath, "cache_size": 256, "model_seed": 3, "return_tensors": True}
    if raw:
        cfg.update(rawstate_as_obs=True, support_rllib_mask=True)
    return cfg


def _env(cfg, seq=False):
    import rl4rs_amd
    if seq:
        from rl4rs_amd.env.seqslate import SeqSlateRecEnv, SeqSlateState
        return rl4rs_amd.make('SeqSlateRecEnv-v0', recsim=SeqSlateRecEnv(cfg, state_cls=SeqSlateState))
    from rl4rs_amd.env.slate import SlateRecEnv, SlateState
    return rl4rs_amd.make('SlateRecEnv-v0', recsim=SlateRecEnv(cfg, state_cls=SlateState))


def _init_dist(rank, world, port):
    import torch
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    from rl4rs_amd import dist as D
    D.init('gloo')
    return D


def _trainer_worker(rank, world, port, d, algo, iters, out):
    import torch
    D = _init_dist(rank, world, port)
    from rl4rs_amd.train import Trainer
    env = _env(_make_cfg(d, rank))
    env.seed(100 + rank)
    tr = Trainer(env, algo=algo, seed=1 + rank, init_seed=5, lr=1e-3, minibatch=128, keep_last_batch=True)
    p0 = tr.params().cpu()
    log = []
    for i in range(iters):
        st = tr.train_iteration()
        lb = dict((k, (v.cpu() if torch.is_tensor(v) else v)) for k, v in tr.last_batch.items())
        log.append(dict(params=tr.params().cpu(), batch=lb, stats=st))
    D.barrier()
    torch.save(dict(p0=p0, log=log), os.path.join(d, 'rank%d.pt' % rank))
    out.put(rank)


def _spawn(target, args, world=2, deadline=420):
    """Run ``target(rank, world, port, *args, queue)`` in ``world`` spawned processes; fail fast when one dies (a dead rank
    would otherwise leave its peer blocked in a collective until the box's limit)."""
    import queue as pyqueue
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    done, t0 = [], time.time()
    try:
        while len(done) < world:
            try:
                done.append(q.get(timeout=2))
            except pyqueue.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                assert not dead, 'a rank exited with %r' % (dead,)
                assert time.time() - t0 < deadline, 'ranks did not finish within %d s' % deadline
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert sorted(done) == list(range(world))


@pytest.mark.parametrize('algo', ['A2C', 'PPO'])
def test_trainer_two_ranks_on_one_gpu(tmp_path, algo):
    import torch
    from rl4rs_amd.device import DevicePolicy
    d = str(tmp_path)
    iters = 2
    _spawn(_trainer_worker, (d, algo, iters))
    r0 = torch.load(os.path.join(d, 'rank0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(d, 'rank1.pt'), weights_only=False)
    assert torch.equal(r0['p0'], r1['p0'])                       # replicas start identical (shared init + broadcast)
    # the ranks sampled DIFFERENT rollouts ...
    assert not torch.equal(r0['log'][0]['batch']['obs'], r1['log'][0]['batch']['obs'])
    assert not torch.equal(r0['log'][0]['batch']['act'], r1['log'][0]['batch']['act'])
    # ... single-process run on the mean of the two shard gradients
    pol = DevicePolicy(256, 64, 284, max_rows=64 * 9, params=r0['p0'].numpy())
    c = lambda t: t.cuda().contiguous()
    for i in range(iters):
        a, b = r0['log'][i], r1['log'][i]
        assert torch.equal(a['params'], b['params']), 'iteration %d: replicas diverged' % i
        assert a['stats']['kl_coeff'] == b['stats']['kl_coeff']
        ba, bb = a['batch'], b['batch']
        if algo == 'A2C':
            gs = []
            for bt in (ba, bb):
                g, _ = pol.loss_grad(0, c(bt['obs']), c(bt['act']), c(bt['adv']), c(bt['ret']), mask_bits=c(bt['mask']),
                                     vf_coeff=0.5, ent_coeff=0.01)
                gs.append(g.clone())
            pol.adam_step((gs[0] + gs[1]) / 2, lr=1e-3, grad_clip=10.0)
        else:
            assert ba['kl_coeff'] == bb['kl_coeff']
            N, MB = ba['obs'].shape[0], 128
            dev = [dict((k, c(v)) for k, v in bt.items() if torch.is_tensor(v)) for bt in (ba, bb)]
            for mb in range(N // MB):
                gs = []
                for bt in dev:
                    g, _ = pol.ppo_minibatch_grad(mb, bt['obs'], bt['act'], bt['adv'], bt['ret'], bt['mask'], bt['logp'], bt['val'],
                                                  bt['logits'], minibatch=MB, kl_coeff=ba['kl_coeff'])
                    gs.append(g.clone())
                pol.adam_step((gs[0] + gs[1]) / 2, lr=1e-3)
        ref = pol.params().cpu()
        assert torch.isfinite(ref).all() and not torch.equal(ref, r0['p0'])
        assert (a['params'] - ref).abs().max().item() <= 1e-6, (i, (a['params'] - ref).abs().max().item())


def test_ppo_minibatch_grad_matches_pass_and_chain():
    """rl4rs_policy_ppo_minibatch_grad (phase A + gradient tiles of k_ppo_pass as one launch, no update) followed by
    rl4rs_policy_adam_step == the fused single-GPU pass, bit for bit; and its gradient == the per-minibatch kernel chain's
    within rounding.  Also: the pass reports the pass-wide loss sums (stats[4:8])."""
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    rs = np.random.RandomState(3)
    N, MB, A = 512, 128, 284
    flat = init_policy_params(256, 64, A, 2)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    obs = t(rs.normal(size=(N, 256)).astype(np.float32))
    mask = rs.rand(N, A) < 0.6
    mask[:, 0] = True
    bits = np.zeros((N, 9), dtype=np.uint32)
    for j in range(A):
        bits[:, j // 32] |= (mask[:, j].astype(np.uint32) << np.uint32(j % 32))
    bits = t(bits.view(np.int32))
    p1, p2, p3 = (DevicePolicy(256, 64, A, max_rows=N, params=flat) for _ in range(3))
    a, lp, v, ent, lg = p1.act(obs, bits, seed=1, step=0, want_logits=True)
    adv = t(rs.normal(size=N).astype(np.float32))
    ret = t(rs.normal(size=N).astype(np.float32) * 3)
    kw = dict(vf_coeff=0.5, ent_coeff=0.0, clip=0.3, vf_clip=500.0, kl_coeff=0.3)
    p1.set_params(t(flat) * 1.01)
    p2.set_params(t(flat) * 1.01)
    p3.set_params(t(flat) * 1.01)
    s8 = p1.ppo_epoch(obs, a, adv, ret, bits, lp, v, lg, minibatch=MB, lr=1e-3, **kw).cpu().numpy()
    kl_sum = 0.0
    for mb in range(N // MB):
        g, st = p2.ppo_minibatch_grad(mb, obs, a, adv, ret, bits, lp, v, lg, minibatch=MB, **kw)
        lo, hi = mb * MB, (mb + 1) * MB
        g3, st3 = p3.loss_grad(1, obs[lo:hi], a[lo:hi], adv[lo:hi], ret[lo:hi], mask_bits=bits[lo:hi], old_logp=lp[lo:hi],
                               old_value=v[lo:hi], old_logits=lg[lo:hi], **kw)
        assert torch.allclose(g, g3, rtol=2e-4, atol=2e-6)
        assert torch.allclose(st, st3, rtol=1e-4, atol=1e-4)
        kl_sum += float(st[3])
        p2.adam_step(g, lr=1e-3)
        p3.adam_step(g, lr=1e-3)            # keep the chain policy on the same trajectory
    assert torch.equal(p1.params(), p2.params())
    assert np.allclose(s8[7], kl_sum, rtol=1e-4, atol=1e-5)
    assert np.allclose(s8[:4], st.cpu().numpy(), rtol=1e-4, atol=1e-4)
    p1.check_status()
    p2.check_status()


def test_pass_falls_back_when_the_grid_cannot_be_resident():
    """the persistent pass is only used when its whole grid fits the device at once (runtime occupancy x CU count);
    a smaller device (pretended per handle: rl4rs_policy_set_option RESIDENT_WGS) takes the per-minibatch kernels and gives
    the same parameters within rounding.  Also ADVICE r2: two handles of DIFFERENT shapes in one process - the pass kernel's
    dynamic-LDS opt-in is per function, so the smaller policy must not undercut the larger one's launches."""
    import torch
    from rl4rs_amd.device import DevicePolicy
    from rl4rs_amd.nets.policy import init_policy_params
    rs = np.random.RandomState(0)
    N, MB, A = 256, 128, 284
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    obs = t(rs.normal(size=(N, 256)).astype(np.float32))
    adv, ret = t(rs.normal(size=N).astype(np.float32)), t(rs.normal(size=N).astype(np.float32))

    def run(hidden, cap=None, fused=None):
        p = DevicePolicy(256, hidden, A, max_rows=N, params=init_policy_params(256, hidden, A, 2))
        if cap is not None:
            p.set_option('resident_wgs', cap)
        if fused is not None:
            p.set_option('ppo_fused', fused)
        a, lp, v, ent, lg = p.act(obs, None, seed=1, step=0, want_logits=True)
        return p, (a, lp, v, lg)

    big, bi = run(128)                     # larger LDS footprint: opts in first ...
    small, si = run(64)                    # ... then a smaller policy runs its own pass
    capped, ci = run(64, cap=4)
    chain, hi = run(64, fused=0)
    for _ in range(2):                     # interleaved: big - small - big
        for p, (a, lp, v, lg) in ((big, bi), (small, si), (capped, ci), (chain, hi)):
            p.ppo_epoch(obs, a, adv, ret, None, lp, v, lg, minibatch=MB, lr=1e-3)
    for p in (big, small, capped, chain):
        p.check_status()
    ps = [p.params().cpu().numpy() for p in (big, small, capped, chain)]
    assert all(np.isfinite(x).all() for x in ps)
    assert np.abs(ps[1] - ps[2]).max() < 5e-6 and np.abs(ps[1] - ps[3]).max() < 5e-6
    big2, b2 = run(128, fused=0)           # the larger policy's pass really ran: equal to its own kernel chain
    for _ in range(2):
        big2.ppo_epoch(obs, b2[0], adv, ret, None, b2[1], b2[2], b2[3], minibatch=MB, lr=1e-3)
    assert np.abs(ps[0] - big2.params().cpu().numpy()).max() < 5e-6
    with pytest.raises(ValueError):
        small.set_option('no_such_option', 1)


def test_trainer_tracks_fp64_ppo_restatement(tmp_path):
    """Three PPO train calls: device parameters, pass-mean KL and the adaptive kl_coeff against the float64 restatement
    (oracle/policy.py ppo_train_call), teacher-forced on the device's own rollouts."""
    import torch
    from rl4rs_amd.train import Trainer
    from oracle import policy as OP
    env = _env(_make_cfg(str(tmp_path), 0))
    env.seed(7)
    tr = Trainer(env, algo='PPO', seed=3, init_seed=9, lr=1e-3, minibatch=128, keep_last_batch=True, kl_target=0.01)
    flat = tr.params().cpu().numpy().astype(np.float64)
    state = (flat, np.zeros_like(flat), np.zeros_like(flat), 0)
    kl_coeff = 0.2

    def unpack(bits):
        b = bits.view(np.uint32)
        return ((b[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(b.shape[0], -1)[:, :284].astype(np.float64)

    coeffs = []
    for it in range(3):
        st = tr.train_iteration()
        lb = tr.last_batch
        assert lb['kl_coeff'] == kl_coeff
        batch = dict((k, lb[k].cpu().numpy()) for k in ('obs', 'act', 'mask', 'adv', 'ret', 'logp', 'val', 'logits'))
        state, ref = OP.ppo_train_call(state, batch, 128, 1e-3, kl_coeff, 0.01, unpack)
        got = tr.params().cpu().numpy()
        assert np.abs(got - state[0]).max() < 2e-5, (it, np.abs(got - state[0]).max())
        assert abs(st['kl_mean'] - ref['kl_mean']) < 1e-5 + 1e-3 * abs(ref['kl_mean'])
        assert np.allclose([st['policy_loss'] / 128, st['vf_loss'] / 128, st['entropy'] / 128, st['kl'] / 128], ref['last'],
                           rtol=2e-3, atol=1e-4)
        kl_coeff = ref['kl_coeff']
        assert st['kl_coeff'] == kl_coeff
        coeffs.append(kl_coeff)
    assert coeffs[0] != 0.2                 # the rule fired (lr 1e-3 moves the policy: KL leaves [target/2, 2 target])


def _raw_worker(rank, world, port, d, algo, out):
    import torch
    D = _init_dist(rank, world, port)
    from rl4rs_amd.train import RawStateTrainer
    env = _env(_make_cfg(d, rank, B=32, raw=True))
    env.seed(100 + rank)
    tr = RawStateTrainer(env, algo=algo, seed=1 + rank, init_seed=5, lr=1e-3, minibatch=96)
    assert tr._grad is not None and tr._grad.data_ptr() == tr.policy.flat_view('grad').data_ptr()      # zero-copy alias
    p0 = tr.policy._flat('params').cpu()
    st = tr.train_iteration()
    p1 = tr.policy._flat('params').cpu()
    D.barrier()
    torch.save(dict(p0=p0, p1=p1, stats=st), os.path.join(d, 'raw%d.pt' % rank))
    out.put(rank)


@pytest.mark.parametrize('algo', ['A2C', 'PPO'])
def test_rawstate_trainer_two_ranks_on_one_gpu(tmp_path, algo):
    import torch
    d = str(tmp_path)
    _spawn(_raw_worker, (d, algo))
    r0 = torch.load(os.path.join(d, 'raw0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(d, 'raw1.pt'), weights_only=False)
    assert torch.equal(r0['p0'], r1['p0'])
    assert torch.equal(r0['p1'], r1['p1']), 'raw-state replicas diverged'
    assert torch.isfinite(r0['p1']).all() and not torch.equal(r0['p0'], r0['p1'])
    assert r0['stats']['kl_coeff'] == r1['stats']['kl_coeff']


def test_sparse_row_allreduce_equals_dense_mean():
    """dist.allreduce_rows_mean_ on this process' own 'world of one' is the identity; the 2-rank form is covered on CPU
    (tests/test_dist_gloo.py) and through the raw-state trainer above."""
    import torch
    from rl4rs_amd import dist as D
    g = torch.randn(100, 8, device='cuda')
    ids = torch.tensor([3, 7], device='cuda')
    ref = g.clone()
    assert torch.equal(D.allreduce_rows_mean_(g, ids), ref)


def _bc_worker(rank, world, port, d, out):
    import torch
    D = _init_dist(rank, world, port)
    from rl4rs_amd.offline_rl import DiscreteBC
    cfg = _make_cfg(d, 0)
    data = torch.load(os.path.join(d, 'bc_data.pt'), weights_only=False)
    obs, act = data['obs'][rank].cuda(), data['act'][rank].cuda()
    bc = DiscreteBC(cfg, obs.shape[1], batch_size=obs.shape[0] // 2, learning_rate=1e-3, seed=4)
    for k in range(2):
        lo, hi = k * bc.batch_size, (k + 1) * bc.batch_size
        bc.update(obs[lo:hi].contiguous(), act[lo:hi].contiguous())
    w = dict((k, v.cpu()) for k, v in bc.imitator.weights().items())
    D.barrier()
    torch.save(w, os.path.join(d, 'bc%d.pt' % rank))
    out.put(rank)


def test_offline_learner_two_ranks_on_one_gpu(tmp_path):
    """offline_rl._Learner on its real device network at world_size 2: replicas identical, and equal to ONE process training
    on the concatenated minibatches (the imitation loss is a batch mean, so the mean of two half-batch gradients is the
    full-batch gradient)."""
    import torch
    from rl4rs_amd.offline_rl import DiscreteBC
    d = str(tmp_path)
    cfg = _make_cfg(d, 0)
    rs = np.random.RandomState(0)
    n = 64

    def shard():
        obs = rs.normal(size=(n, 266)).astype(np.float32)
        prev = np.stack([rs.choice(np.arange(1, 284), size=9, replace=False) for _ in range(n)])
        step = rs.randint(0, 9, size=n)
        for i in range(n):
            prev[i, step[i]:] = 0
        obs[:, 256:265] = prev
        obs[:, 265] = step
        return torch.from_numpy(obs), torch.from_numpy(rs.randint(1, 284, size=n).astype(np.int32))

    s0, s1 = shard(), shard()
    torch.save(dict(obs=[s0[0], s1[0]], act=[s0[1], s1[1]]), os.path.join(d, 'bc_data.pt'))
    _spawn(_bc_worker, (d,))
    w0 = torch.load(os.path.join(d, 'bc0.pt'), weights_only=False)
    w1 = torch.load(os.path.join(d, 'bc1.pt'), weights_only=False)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
    bc = DiscreteBC(cfg, 266, batch_size=n, learning_rate=1e-3, seed=4)
    h = n // 2
    for k in range(2):
        obs = torch.cat([s0[0][k * h:(k + 1) * h], s1[0][k * h:(k + 1) * h]]).cuda().contiguous()
        act = torch.cat([s0[1][k * h:(k + 1) * h], s1[1][k * h:(k + 1) * h]]).cuda().contiguous()
        bc.update(obs, act)
    ref = bc.imitator.weights()
    for k in w0:
        # Adam divides by sqrt(v): rounding differences of near-zero gradients are amplified up to ~lr * 1e-2
        assert (w0[k] - ref[k].cpu()).abs().max().item() < 2e-5, k
