#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace into a markdown table:
per kernel (and per grid size): calls, total / avg / min / max duration, share.

usage: python tools/rocpd_summary.py <results.db> [--by-grid] > profiles/<name>.md
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = '--by-grid' in sys.argv
    key = "name, grid_x, grid_y, workgroup_x" if by_grid else "name"
    rows = db.execute(
        "select %s, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by %s "
        "order by sum(duration) desc" % (key, key)).fetchall()
    total = sum(r[-7] for r in rows) or 1
    hdr = (['kernel', 'grid_x', 'grid_y', 'wg_x'] if by_grid else ['kernel']) + \
        ['calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'agpr', 'lds_B']
    print('| ' + ' | '.join(hdr) + ' |')
    print('|' + '---|' * len(hdr))
    for r in rows:
        head = list(r[:-8])
        calls, tot, avg, mn, mx, vg, ag, lds = r[-8:]
        name = str(head[0])
        if len(name) > 90:
            name = name[:87] + '...'
        head[0] = '`' + name + '`'
        vals = head + [calls, '%.3f' % (tot / 1e6), '%.1f' % (avg / 1e3), '%.1f' % (mn / 1e3), '%.1f' % (mx / 1e3),
                       '%.1f' % (100.0 * tot / total), vg, ag, lds]
        print('| ' + ' | '.join(str(v) for v in vals) + ' |')


if __name__ == '__main__':
    main()
