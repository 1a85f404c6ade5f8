"""cpu_baseline pool geometry sweep on the bench box: (workers, rows per worker) -> env-steps/s of the vectorised CPU port
(oracle/cpu_pool.py).  usage: python tools/cpu_pool_sweep.py 64x64 128x32 256x16"""
import sys
import tempfile

sys.path.insert(0, '.')

if __name__ == '__main__':
    import bench
    from oracle.cpu_pool import run_pool

    class A(object):
        env, batch, horizon, log_records, scorer, algo, conti = 'slate', 4096, 9, 8193, 'auto', 'dien', False

    cfg, records = bench.make_config(A(), tempfile.mkdtemp(), 0)
    for spec in sys.argv[1:] or ['128x32']:
        w, r = [int(x) for x in spec.split('x')]
        res = run_pool(dict(cfg), records[:w * r], False, w, r)
        print('%s: %.0f env-steps/s (%d envs, %.1f s wall, slowest worker %.1f s)' % (spec, res['env_steps'] / res['seconds'], w * r, res['seconds'], res['slowest_worker_s']), flush=True)
