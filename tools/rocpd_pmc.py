#!/usr/bin/env python
"""Per-kernel PMC summary from a rocprofv3 rocpd SQLite db: avg counter value per dispatch, grouped by
kernel name and grid size.  usage: python tools/rocpd_pmc.py <results.db>"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info('pmc_events')")]
    q = ("select p.name, k.grid_x, k.grid_y, p.counter_name, count(*), avg(p.counter_value), avg(p.duration) "
         "from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
         "group by p.name, k.grid_x, k.grid_y, p.counter_name order by sum(p.duration) desc")
    try:
        rows = db.execute(q).fetchall()
    except Exception as e:
        print('columns of pmc_events:', cols)
        kc = [r[1] for r in db.execute("pragma table_info('kernels')")]
        print('columns of kernels:', kc)
        raise
    print('| kernel | grid_x | grid_y | counter | dispatches | avg_value | avg_us |')
    print('|---|---|---|---|---|---|---|')
    for name, gx, gy, cn, n, v, d in rows:
        name = name if len(name) < 70 else name[:67] + '...'
        print('| `%s` | %d | %d | %s | %d | %.4g | %.1f |' % (name, gx, gy, cn, n, v, d / 1e3))


if __name__ == '__main__':
    main()
