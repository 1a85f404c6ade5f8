"""cpu_baseline pool geometry sweep on the bench box: (workers, rows per worker) -> env-steps/s of the vectorised CPU port
(oracle/cpu_pool.py).  usage: python tools/cpu_pool_sweep.py 64x64 128x32 256x16 32x128x4   (workers x rows [x threads per worker])"""
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
        parts = [int(x) for x in spec.split('x')]
        w, r, th = parts[0], parts[1], (parts[2] if len(parts) > 2 else 1)
        res = run_pool(dict(cfg), records[:w * r], False, w, r, threads=th)
        print('%s: %.0f env-steps/s (%d envs, %.1f s wall, slowest worker %.1f s)' % (spec, res['env_steps'] / res['seconds'], w * r, res['seconds'], res['slowest_worker_s']), flush=True)
