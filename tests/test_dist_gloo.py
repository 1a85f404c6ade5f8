"""CPU: the N>1 plumbing with world_size 2 over gloo (sharding, max-over-ranks timing, gradient mean)."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from rl4rs_amd import dist as D
    r, lr, w = D.init('gloo')
    assert (r, w) == (rank, world)
    lo, hi = D.shard_rows(4097, r, w)
    t = D.max_over_ranks(1.0 + r)
    total = D.sum_over_ranks(hi - lo)
    g = torch.full((34973,), float(r + 1))
    D.allreduce_mean_(g)
    D.barrier()
    out.put((r, lo, hi, t, total, float(g[0]), float(g[-1]), D.shard_seed(1000, r)))


def test_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1:3] for r in res] == [(0, 2049), (2049, 4097)]          # contiguous, covering, balanced
    assert all(r[3] == 2.0 for r in res)                               # MAX over ranks
    assert all(r[4] == 4097.0 for r in res)                            # every env row owned exactly once
    assert all(r[5] == 1.5 and r[6] == 1.5 for r in res)               # gradient mean
    assert [r[7] for r in res] == [1000, 1001]


def test_single_process_is_a_noop():
    from rl4rs_amd import dist as D
    import torch
    assert D.shard_rows(10, 0, 1) == (0, 10)
    assert D.max_over_ranks(3.5) == 3.5
    g = torch.ones(4)
    assert D.allreduce_mean_(g) is g


class _StubNet(object):
    """Stands in for a DeviceQNet: a flat gradient buffer and a plain SGD 'adam_step' so the update is checkable on CPU."""

    def __init__(self, grad):
        import torch
        self.g = grad.clone()
        self.p = torch.zeros_like(grad)
        self.steps = 0

    def flat_gradient(self):
        return self.g.clone()

    def set_flat_gradient(self, g):
        self.g = g.clone()

    def adam_step(self, lr):
        self.p -= lr * self.g
        self.steps += 1


def _learner_worker(rank, world, port, out):
    import torch
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from rl4rs_amd import dist as D
    from rl4rs_amd.offline_rl import _Learner
    D.init('gloo')
    learner = _Learner.__new__(_Learner)           # the data-parallel update rule only: no device network is built
    learner.lr = 0.5
    net = _StubNet(torch.arange(6, dtype=torch.float32) * (rank + 1))
    learner._apply(net)
    D.barrier()
    out.put((rank, net.g.tolist(), net.p.tolist(), net.steps))


def test_offline_learner_update_is_data_parallel():
    """offline_rl._Learner._apply: each rank's flat gradient is mean-all-reduced before the optimiser step (SURVEY 8e:
    'BCQ/offline: data-parallel minibatches, gradient all-reduce only'), so every rank applies the same update."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_learner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean = [1.5 * k for k in range(6)]                                   # ranks hold k and 2k
    for _, g, p, steps in res:
        assert g == mean and p == [-0.5 * x for x in mean] and steps == 1


def _group_worker(rank, world, port, out):
    import torch
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from rl4rs_amd import dist as D
    from rl4rs_amd.offline_rl import _allreduce_group
    D.init('gloo')
    nets = [_StubNet(torch.arange(4, dtype=torch.float32) * (rank + 1)), _StubNet(torch.ones(3) * (10 * rank))]
    _allreduce_group(nets)
    D.barrier()
    out.put((rank, D.rank(), nets[0].g.tolist(), nets[1].g.tolist()))


def test_continuous_learner_groups_its_gradient_allreduce():
    """offline_rl._allreduce_group (continuous BCQ / CQL): the flat gradients of the networks stepped together (the two halves
    of the VAE, the twin critics) travel as ONE mean all-reduce and come back split per network."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, drank, g0, g1 in res:
        assert rank == drank
        assert g0 == [1.5 * k for k in range(4)] and g1 == [5.0] * 3


def _rows_worker(rank, world, port, out):
    import torch
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from rl4rs_amd import dist as D
    D.init('gloo')
    g = torch.Generator().manual_seed(10 + rank)
    H, E = 1000, 8
    ids = torch.unique(torch.randint(0, H, (40,), generator=g))
    table = torch.zeros(H, E)
    table[ids] = torch.randn(ids.numel(), E, generator=g)
    dense = table.clone()
    D.allreduce_mean_(dense.view(-1))                      # what the sparse-row exchange must reproduce
    # the ids as a trainer hands them over: every id slot of the minibatch, repeats included, unsorted; one rank touches row 0
    slots = torch.cat([ids, ids[:7], ids[3:5]])[torch.randperm(ids.numel() + 9, generator=g)]
    if rank == 1:
        table[0] = 3.0
        dense2 = torch.zeros(H, E); dense2[0] = 3.0
        slots = torch.cat([slots, torch.zeros(2, dtype=torch.int64)])
    else:
        dense2 = torch.zeros(H, E)
    D.allreduce_mean_(dense2.view(-1))
    dense += dense2
    D.allreduce_rows_mean_(table, slots, cap=64)           # buffer sizes must agree across ranks: the slot counts differ here
    assert not bool(D.take_row_overflow(torch.device('cpu'))) and D.take_row_overflow('cpu') is None       # read once, then reset
    # a distinct-row hint that is too small must fail loudly (NaN rows), never drop rows silently
    bad = torch.zeros(H, E)
    bad[ids] = 1.0
    D.allreduce_rows_mean_(bad, ids, cap=3)
    assert torch.isnan(bad).any()
    assert bool(D.take_row_overflow('cpu'))                # ... and the flag the trainers fold into their status word is set
    # only ONE rank overflows: the flag is global (its bit travels with the ids), so every rank raises at the same settle
    lop = torch.zeros(H, E)
    lop_ids = ids[:12] if rank == 0 else ids[:3]
    lop[lop_ids] = 1.0
    D.allreduce_rows_mean_(lop, lop_ids, cap=5)
    assert torch.isnan(lop).any() and bool(D.take_row_overflow('cpu')), rank
    big = torch.zeros(16, 4)                               # touched rows are most of the table -> dense fallback
    big_ids = torch.arange(rank, 16, 2)
    big[big_ids] = float(rank + 1)
    D.allreduce_rows_mean_(big, big_ids)
    assert D.LAST_ROWS_PATH == 'dense'
    # the bench shape of the raw-state policy's SEQUENCE table: 256 x 64 x 2 = 32768 id slots naming at most 283
    # distinct item ids of a 100000-row table.  With the slot count as the bound every W >= 2 took the dense 51 MB all-reduce;
    # the calibrated bound keeps it on the sparse exchange
    Hs, n_slots = 100000, 256 * 64 * 2
    seq_ids = torch.randint(1, 284, (n_slots,), generator=g)
    seq_tab = torch.zeros(Hs, 4)
    seq_tab[torch.unique(seq_ids)] = float(rank + 1)
    cap = D.calibrate_row_cap(seq_ids, Hs)
    assert cap == 2048 and cap * 2 * 2 <= Hs
    D.allreduce_rows_mean_(seq_tab, seq_ids, cap=cap)
    assert D.LAST_ROWS_PATH == 'sparse'
    touched = torch.zeros(Hs, dtype=torch.bool)
    touched[1:284] = True
    assert (seq_tab[~touched] == 0).all() and not torch.isnan(seq_tab).any() and (seq_tab[touched] > 0).all()
    D.allreduce_rows_mean_(torch.zeros(Hs, 4), seq_ids)          # without the hint: the id-slot count -> dense fallback
    assert D.LAST_ROWS_PATH == 'dense'
    b = torch.full((5,), float(rank))
    D.broadcast_(b, src=1)
    D.barrier()
    out.put((rank, table.numpy().copy(), dense.numpy().copy(), big.numpy().copy(), b.numpy().copy()))


def test_sparse_row_allreduce_and_broadcast_world_2():
    """dist.allreduce_rows_mean_ (embedding-table gradients of the raw-state policy: only touched rows travel) == the dense
    mean all-reduce, bit-identical on both ranks; dist.broadcast_."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, t0, d0, b0, bc0), (_, t1, d1, b1, bc1) = [tuple(torch.from_numpy(x) if hasattr(x, 'shape') else x for x in r) for r in res]
    assert torch.equal(t0, t1) and torch.equal(d0, d1)
    assert torch.equal(t0, d0)                             # W = 2: x0/2 + x1/2 == (x0 + x1)/2 exactly
    assert (t0 != 0).any()
    assert torch.equal(b0, b1) and set(b0.unique().tolist()) == {0.5, 1.0}
    assert torch.equal(bc0, torch.ones(5)) and torch.equal(bc1, torch.ones(5))


def _w8_worker(rank, world, port, out):
    import torch
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from rl4rs_amd import dist as D
    r, lr, w = D.init('gloo')
    assert (r, w) == (rank, world)
    lo, hi = D.shard_rows(32768, r, w)
    floats = D.gather_floats(10.0 * r + 0.5)                  # rank order on every rank
    t = D.max_over_ranks(1.0 + r)
    total = D.sum_over_ranks(hi - lo)
    g = torch.full((34973,), float(r + 1))
    D.allreduce_mean_(g)
    # sparse row exchange at W = 8: every rank touches its own rows plus a few shared ones (row 0 among them)
    gen = torch.Generator().manual_seed(100 + rank)
    H, E = 4000, 8
    ids = torch.unique(torch.cat([torch.randint(0, H, (30,), generator=gen), torch.tensor([0, 7, 8, 9])]))
    table = torch.zeros(H, E)
    table[ids] = torch.randn(ids.numel(), E, generator=gen)
    dense = table.clone()
    D.allreduce_mean_(dense.view(-1))
    slots = torch.cat([ids, ids[:5]])[torch.randperm(ids.numel() + 5, generator=gen)]
    D.allreduce_rows_mean_(table, slots, cap=64)
    path = D.LAST_ROWS_PATH
    over = D.take_row_overflow('cpu')
    b = torch.full((3,), float(rank))
    D.broadcast_(b, src=5)
    D.barrier()
    out.put((rank, lo, hi, floats, t, total, float(g[0]), table.numpy().copy(), dense.numpy().copy(), path, bool(over), b.numpy().copy()))


def test_world_size_8_gloo():
    """Nothing with more than two ranks had run before round 6: the wrappers bench.py and the trainers use, at the
    driver's 8-rank world size - contiguous balanced row blocks of the 32768-env configuration, gather order = rank order, the
    sparse row exchange bit-identical ACROSS ranks and equal to the dense mean to float32 rounding (eight addends in a fixed rank
    order against gloo's reduction tree), broadcast from a non-zero root."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    W = 8
    procs = [ctx.Process(target=_w8_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(4096 * k, 4096 * (k + 1)) for k in range(W)]          # configs[3]: 8 x 4096 envs
    for r in res:
        assert r[3] == [10.0 * k + 0.5 for k in range(W)]
        assert r[4] == float(W) and r[5] == 32768.0 and r[6] == 4.5
        assert r[9] == 'sparse' and r[10] is False
        assert (r[11] == 5.0).all()
    t0, d0 = torch.from_numpy(res[0][7]), torch.from_numpy(res[0][8])
    for r in res[1:]:
        assert torch.equal(torch.from_numpy(r[7]), t0)                 # bit-identical on every rank
    assert torch.allclose(t0, d0, rtol=1e-6, atol=1e-7) and (t0 != 0).any()


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """`bench.py --gpus N` under a launcher whose WORLD_SIZE is not N exits non-zero before anything runs (the entry
    point used to report n_gpus = 1 for `--gpus 8` without a launcher; now it starts its own ranks, and a mismatch is an error)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2'], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0 and b'WORLD_SIZE=4 but --gpus 2' in out.stderr and not out.stdout.strip()


def test_forced_collectives_on_a_one_rank_group():
    """RL4RS_DIST_FORCE / set_force: a ONE-rank group still enters every collective (the switch the one-rank RCCL GPU test uses);
    here over gloo on CPU: results are the identity."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RL4RS_DIST_FORCE='1')
from rl4rs_amd import dist as D
import torch.distributed as dist
assert D.FORCE_COLLECTIVES and not D.collectives_active()
assert D.init('gloo') == (0, 0, 1) and dist.is_initialized() and D.collectives_active()
calls = []
real = dist.all_reduce
dist.all_reduce = lambda *a, **k: (calls.append('ar'), real(*a, **k))[1]
g = torch.arange(10, dtype=torch.float32)
assert torch.equal(D.allreduce_mean_(g.clone()), g) and calls == ['ar']
assert D.max_over_ranks(2.5) == 2.5 and D.sum_over_ranks(1.0) == 1.0 and D.gather_floats(4.0) == [4.0]
assert len(calls) == 3
t = torch.zeros(5000, 4); ids = torch.tensor([5, 9, 5]); t[ids] = 1.0
w = t.clone()
D.allreduce_rows_mean_(t, ids, cap=8)
assert D.LAST_ROWS_PATH == 'sparse' and torch.equal(t, w)
D.set_force(False)
assert not D.collectives_active()
assert D.allreduce_mean_(g) is g and len(calls) == 3
D.barrier()
print('OK')
''' % (repo, _free_port())
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0 and b'OK' in out.stdout, out.stderr.decode()[-2000:]
