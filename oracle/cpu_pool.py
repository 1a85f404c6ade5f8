"""Row-block process pool for bench.py's ``cpu_baseline``: the vectorised CPU port on ALL host cores.
ORACLE - test infrastructure only (see oracle/__init__.py); nothing under rl4rs_amd/ imports this.

Env rows are independent (rl4rs/env/base.py:157-170 has no cross-row term), so the sample batch is cut into row blocks and
every block runs the whole episode - numpy state machine (oracle/env.py) + torch-CPU float32 DIEN (oracle/dien_torch.py) -
in its own single-threaded process: no GIL, no shared thread pool, one core per block.  Workers are SPAWNED (a forked torch
process can hang in its inherited OpenMP pool), build their env, report ready, wait for the start signal and only then run the timed episode; the
parent times start signal -> last worker done, i.e. the wall time of the whole sample on the whole machine."""
import os
import time


def _worker(idx, cfg, records, seq, algo, go, out_q, threads=1):
    os.environ['OMP_NUM_THREADS'] = str(threads)
    os.environ['MKL_NUM_THREADS'] = str(threads)
    os.environ['OPENBLAS_NUM_THREADS'] = str(threads)
    try:
        import numpy as np
        import torch
        torch.set_num_threads(threads)
        from oracle.env import OracleEnv
        if algo == 'dien':
            from rl4rs_amd.nets.dien import init_dien_weights
            from oracle.dien_torch import TorchDien
            scorer = TorchDien(init_dien_weights(cfg, seed=cfg.get('model_seed', 7)), cfg, workers=threads)     # row-parallel threads inside the block
        else:
            from rl4rs_amd.nets.simnets import init_simnet_weights
            from oracle.simnets import OracleSimnet
            scorer = OracleSimnet(algo, init_simnet_weights(cfg, algo, seed=cfg.get('model_seed', 7)), cfg, np.float32)
        env = OracleEnv(cfg, records, scorer, seq=seq)
        T = cfg['max_steps']
        out_q.put((idx, 'ready'))
        if not go.wait(timeout=900):
            raise RuntimeError('start signal never came')
        t0 = time.time()
        env.reset()
        for _ in range(T):
            env.step(np.asarray(env.samples.offline_action))
        out_q.put((idx, time.time() - t0, len(records) * T, None))
    except Exception as e:                 # the parent must never wait for a worker that died
        out_q.put((idx, 0.0, 0, repr(e)))


def run_pool(cfg, records, seq, workers, rows_per_worker, threads=1):
    """One episode-batch of ``workers * rows_per_worker`` envs over ``workers`` single-threaded processes.
    -> dict(env_steps, seconds, workers, rows_per_worker, slowest_worker_s)."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    algo = cfg.get('algo', 'dien')
    n = workers * rows_per_worker
    assert len(records) >= n, (len(records), n)
    go = ctx.Event()
    q = ctx.Queue()
    procs = []
    for i in range(workers):
        c = dict(cfg, batch_size=rows_per_worker)
        p = ctx.Process(target=_worker, args=(i, c, records[i * rows_per_worker:(i + 1) * rows_per_worker], seq, algo, go, q, threads),
                        daemon=True)
        p.start()
        procs.append(p)

    def collect(n, what):
        """n messages from the workers; a worker that exits without reporting fails the run instead of hanging it"""
        import queue
        got, deadline = [], time.time() + 900
        while len(got) < n:
            try:
                got.append(q.get(timeout=1.0))
            except queue.Empty:
                dead = [p.pid for p in procs if p.exitcode not in (None, 0)]
                if dead or time.time() > deadline:
                    raise RuntimeError('cpu_pool: %s while waiting for %s (%d of %d reported)'
                                       % ('worker(s) %s died' % dead if dead else 'timeout', what, len(got), n))
                continue
            if len(got[-1]) == 4 and got[-1][3]:
                raise RuntimeError('cpu_pool worker failed: %s' % got[-1][3])
        return got

    try:
        collect(workers, 'the workers to build their envs')
        go.set()
        t0 = time.time()
        res = collect(workers, 'the timed episode')
        dt = time.time() - t0
    finally:
        go.set()
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
    return {'env_steps': sum(r[2] for r in res), 'seconds': dt, 'workers': workers, 'rows_per_worker': rows_per_worker, 'threads': threads,
            'slowest_worker_s': max(r[1] for r in res)}
