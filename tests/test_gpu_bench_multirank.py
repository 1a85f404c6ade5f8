"""N > 1 control flow of bench.py (barriers, max-over-ranks timing, rank-0 reporting, the gradient all-reduce of the training
legs) dry-run with TWO ranks sharing the one GPU of the test box over gloo (RL4RS_DIST_BACKEND=gloo): every leg the driver may
launch at N = 2 / 4 / 8 over RCCL is executed here with the same command line shape, under a hard timeout - a rank waiting in a
collective the others never enter (the round-2 bug of `--train` at N > 1) fails the test instead of hanging the round."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('leg', [[], ['--train', 'a2c'], ['--train', 'ppo'], ['--env', 'seq', '--horizon', '18'], ['--conti'],
                                 ['--train', 'ppo', '--minibatch', '512'], ['--env', 'seq', '--train', 'a2c'],
                                 ['--train', 'bcq', '--bcq-updates', '2']])
def test_bench_two_ranks_over_gloo(leg):
    env = dict(os.environ, RL4RS_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--batch', '256', '--log-records', '700', '--no-cpu-baseline'] + leg
    out = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]              # ONE line, from rank 0 only
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['value'] > 0 and rec['scaling'] == 'weak'
    assert rec['unit'] == 'env-steps/s' and rec['roofline'] is not None
    # what the collective layer saw: both ranks joined the one process group, and each reports its own rate
    assert rec['ranks_seen'] == 2 and len(rec['per_rank_env_steps_per_s']) == 2 and min(rec['per_rank_env_steps_per_s']) > 0
    assert rec['value'] <= sum(rec['per_rank_env_steps_per_s']) * 1.001
    if '--train' in leg:
        assert rec['replicas_in_step'] is True, rec['param_digest_per_rank']
    if leg[:2] == ['--train', 'bcq']:
        assert rec['bcq']['updates_per_step'] == 2 and all(v == v for v in rec['bcq']['last_losses'].values())


def test_plain_launch_starts_its_own_ranks():
    """``python bench.py --gpus 2`` with NO launcher around it (the shape of the driver's one known command): bench.py starts the
    two ranks itself; the line says n_gpus == ranks_seen == 2."""
    env = dict((k, v) for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'))
    env.update(RL4RS_DIST_BACKEND='gloo')
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '256',
           '--log-records', '700', '--no-cpu-baseline']
    out = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2 and len(rec['per_rank_env_steps_per_s']) == 2


@pytest.mark.parametrize('leg', [[], ['--train', 'a2c'], ['--train', 'ppo'], ['--train', 'bcq', '--bcq-updates', '2']])
def test_plain_launch_eight_ranks(leg):
    """The driver's 8-GPU command line shape, dry: plain ``python bench.py --gpus 8`` (no launcher) starts EIGHT ranks that share the
    one GPU over gloo - nothing with more than two ranks had ever run.  All eight join one process group, each reports
    its own rate, and after data-parallel training steps every rank holds the same learner (equal parameter digests)."""
    env = dict((k, v) for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'))
    env.update(RL4RS_DIST_BACKEND='gloo')
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--batch', '256',
           '--log-records', '700', '--no-cpu-baseline'] + leg
    out = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 8 and rec['ranks_seen'] == 8 and rec['scaling'] == 'weak'
    assert len(rec['per_rank_env_steps_per_s']) == 8 and min(rec['per_rank_env_steps_per_s']) > 0
    assert rec['value'] <= sum(rec['per_rank_env_steps_per_s']) * 1.001
    if leg:
        assert len(rec['param_digest_per_rank']) == 8 and rec['replicas_in_step'] is True, rec['param_digest_per_rank']
    else:
        assert rec['param_digest_per_rank'] is None


def test_more_ranks_than_gpus_over_rccl_is_refused():
    """The default backend is one rank per GPU over RCCL: asking for more ranks than the box has GPUs fails loudly, before any
    rank starts, instead of reporting a smaller job."""
    import torch
    n = torch.cuda.device_count() + 1
    env = dict((k, v) for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'RL4RS_DIST_BACKEND'))
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                         cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode != 0 and b'visible GPUs' in out.stderr
    assert not [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
