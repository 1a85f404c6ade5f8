"""GPU: the RCCL ("nccl") branch of every collective wrapper in rl4rs_amd/dist.py, executed for real.

The test box has ONE GPU and RCCL refuses two ranks on one device, so every other multi-rank test goes through gloo's host
staging.  Here a ONE-rank ``nccl`` process group is initialised in a child process with RL4RS_DIST_FORCE=1, which makes the
wrappers enter their collectives although the group has one rank: the device-memory branch (float64 MAX / SUM all-reduce of the
bench timing, all_gather, in-place fp32 SUM all-reduce + division, broadcast, the sparse-row exchange's int64 / fp32 all-gathers
and its dense fall-back, barrier) runs over RCCL on the GPU.  With one rank a mean over ranks is the identity, so:
wrappers return their input exactly, and one ``Trainer.train_iteration()`` (A2C, PPO - the per-minibatch data-parallel path),
one ``RawStateTrainer`` A2C call and ``BCQ.update()`` under the group end on the same parameters as the same calls without a
group.  Reference: none (script/modelfree_train.py:181,349,403 - Ray-internal)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, socket, tempfile
import numpy as np
import torch
sys.path.insert(0, %(repo)r)
sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
from rl4rs_amd import dist as D
import torch.distributed as dist
from test_gpu_train_dp import _make_cfg, _env
from test_gpu_offline_conti import _batch, E, L
from rl4rs_amd.train import Trainer, RawStateTrainer
from rl4rs_amd.offline_rl import BCQ

torch.cuda.set_device(0)
d = tempfile.mkdtemp()


def trainer_params(algo):
    env = _env(_make_cfg(d, 0))
    env.seed(100)
    tr = Trainer(env, algo=algo, seed=1, init_seed=5, lr=1e-3, minibatch=128)
    st = tr.train_iteration()
    kl = st['kl_coeff']
    p = tr.params().cpu()
    tr.close()
    return p, kl


def raw_params():
    env = _env(_make_cfg(d, 0, B=32, raw=True))
    env.seed(100)
    tr = RawStateTrainer(env, algo='A2C', seed=1, init_seed=5, lr=1e-3)
    tr.train_iteration()
    tr._settle()
    return tr.policy._flat('params').cpu()


def bcq_weights():
    B, n = 32, 6
    rs = np.random.RandomState(0)
    f = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32))
    bcq = BCQ({'action_emb_size': E}, 266, batch_size=B, n_action_samples=n, seed=4)
    for k in range(2):
        x, a, rew, ter = _batch(B, 1 + k)
        nx = _batch(B, 101 + k)[0]
        noise = dict(eps=f(rs.randn(B, L)), z_target=f(rs.randn(B * n, L)), z_actor=f(rs.randn(B, L)))
        bcq.update(*[f(v).cuda() for v in (x, a, rew, nx, ter)], noise=noise)
    w = dict((name, dict((k, v.cpu()) for k, v in getattr(bcq, name).weights().items())) for name in ('imit_enc', 'imit_dec', 'policy', 'q1', 'q2', 'q1_targ'))
    bcq.close()
    return w


# ---- without a process group
assert not D.collectives_active()
ref = dict(a2c=trainer_params('A2C'), ppo=trainer_params('PPO'), raw=raw_params(), bcq=bcq_weights())

# ---- a one-rank RCCL group, collectives forced
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
D.set_force(True)
assert D.init('nccl') == (0, 0, 1)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and D.collectives_active() and D.world_size() == 1
dev = torch.device('cuda', 0)
assert D.max_over_ranks(3.25, device=dev) == 3.25
assert D.sum_over_ranks(7.5, device=dev) == 7.5
assert D.gather_floats(1.125, device=dev) == [1.125]
g = torch.randn(34973, device=dev)
g0 = g.clone()
assert D.allreduce_mean_(g) is g and torch.equal(g, g0) and not D._needs_host_staging(g)
t = torch.randn(8, device=dev)
t0 = t.clone()
D.allreduce_sum_(t[7:8])
assert torch.equal(t, t0)
b = torch.arange(1000, device=dev, dtype=torch.float32)
assert torch.equal(D.broadcast_(b.clone(), 0), b)
# sparse-row exchange: 100000 x 16 table, 300 touched rows (duplicates), calibrated cap -> the all_gather branch
H, Ew = 100000, 16
ids = torch.randint(0, H, (300,), device=dev)
ids = torch.cat([ids, ids[:50]])
tg = torch.zeros(H, Ew, device=dev)
tg[ids] = torch.randn(ids.numel(), Ew, device=dev)
want = tg.clone()
cap = D.calibrate_row_cap(ids, H)
assert cap == ids.numel()                    # 4 x ~300 distinct rows, rounded to 2048, bounded by the id-slot count
D.allreduce_rows_mean_(tg, ids, cap=cap)
assert D.LAST_ROWS_PATH == 'sparse' and torch.equal(tg, want)
# the same call on a small table: the dense all-reduce
tg2 = torch.randn(64, Ew, device=dev)
w2 = tg2.clone()
D.allreduce_rows_mean_(tg2, torch.arange(64, device=dev))
assert D.LAST_ROWS_PATH == 'dense' and torch.equal(tg2, w2)
# an overflowing cap is loud (NaN rows), not silent
tg3 = want.clone()
D.allreduce_rows_mean_(tg3, ids, cap=16)
assert torch.isnan(tg3).any() and bool(D.take_row_overflow(dev)) and D.take_row_overflow(dev) is None
D.barrier()

got = dict(a2c=trainer_params('A2C'), ppo=trainer_params('PPO'), raw=raw_params(), bcq=bcq_weights())
for k in ('a2c', 'ppo'):
    assert torch.equal(got[k][0], ref[k][0]), (k, (got[k][0] - ref[k][0]).abs().max().item())
    assert got[k][1] == ref[k][1]
# (the raw-state backward accumulates embedding-row gradients with float atomics: equal up to their summation order)
assert (got['raw'] - ref['raw']).abs().max().item() <= 1e-6 and not torch.equal(ref['raw'], torch.zeros_like(ref['raw']))
for name in ref['bcq']:
    for k in ref['bcq'][name]:
        assert torch.equal(got['bcq'][name][k], ref['bcq'][name][k]), (name, k)
# a sparse exchange whose cap is too small raises at the next settle (ADVICE r4: it used to leave NaN tables behind silently)
env = _env(_make_cfg(d, 0, B=32, raw=True))
env.seed(100)
tr = RawStateTrainer(env, algo='A2C', seed=1, init_seed=5, lr=1e-3)
tr._row_caps = [1, 1]
tr.train_iteration()
try:
    tr._settle()
    raise SystemExit('the row-cap overflow went unnoticed')
except RuntimeError as e:
    assert 'distinct embedding rows' in str(e), str(e)
D.barrier()
dist.destroy_process_group()
print('NCCL_ONE_RANK_OK')
'''


def test_every_wrapper_through_the_rccl_branch():
    env = dict((k, v) for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'RL4RS_DIST_BACKEND'))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.run([sys.executable, '-c', CHILD % dict(repo=REPO)], cwd=REPO, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    if out.returncode != 0:
        pytest.fail('child exited with %d\n%s\n%s' % (out.returncode, out.stdout.decode()[-2000:], out.stderr.decode()[-6000:]), pytrace=False)
    assert 'NCCL_ONE_RANK_OK' in out.stdout.decode()
