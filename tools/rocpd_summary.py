#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace into a markdown table:
per kernel (and per grid size): calls, total / avg / min / max duration, share.

Register columns: `vgpr` / `agpr` / `sgpr` / `spills` come from the CODE OBJECT of the shipped library (tools/codeobj_notes.py:
llvm-readelf --notes; `vgpr` there is the unified total an occupancy argument needs) for every kernel found in it; rocprofv3's
own `vgpr_count` (the granule-rounded arch-VGPR half: 120 / 128 for k_augru_x whose code object says 234 / 254) is only kept,
marked `~`, for kernels that are not ours (torch's).  `lds_B` is the launch's total LDS (static + dynamic) as rocprofv3 saw it.

usage: python tools/rocpd_summary.py <results.db> [--by-grid] [--codeobj path/to/librl4rs_hip.so] > profiles/<name>.md
"""
import os
import sqlite3
import sys


def _codeobj_table():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import codeobj_notes
    so = None
    if '--codeobj' in sys.argv:
        so = sys.argv[sys.argv.index('--codeobj') + 1]
    else:
        so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rl4rs_amd', 'csrc', 'librl4rs_hip.so')
    try:
        return codeobj_notes.kernel_table(so), codeobj_notes.lookup
    except Exception as e:                        # no llvm tools on this box: fall back to rocprofv3's columns, marked
        sys.stderr.write('codeobj metadata unavailable (%s)\n' % e)
        return {}, (lambda t, n: None)


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = '--by-grid' in sys.argv
    co, co_lookup = _codeobj_table()
    key = "name, grid_x, grid_y, workgroup_x" if by_grid else "name"
    rows = db.execute(
        "select %s, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by %s "
        "order by sum(duration) desc" % (key, key)).fetchall()
    total = sum(r[-7] for r in rows) or 1
    hdr = (['kernel', 'grid_x', 'grid_y', 'wg_x'] if by_grid else ['kernel']) + \
        ['calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'agpr', 'sgpr', 'spills', 'lds_B']
    print('| ' + ' | '.join(hdr) + ' |')
    print('|' + '---|' * len(hdr))
    for r in rows:
        head = list(r[:-8])
        calls, tot, avg, mn, mx, vg, ag, lds = r[-8:]
        name = str(head[0])
        if len(name) > 90:
            name = name[:87] + '...'
        e = co_lookup(co, str(head[0]))
        head[0] = '`' + name + '`'
        if e is not None:
            regs = [e.get('vgpr', 0), e.get('agpr', 0), e.get('sgpr', 0), e.get('vgpr_spill', 0) + e.get('sgpr_spill', 0)]
        else:
            regs = ['~%s' % vg, '~%s' % ag, '-', '-']
        vals = head + [calls, '%.3f' % (tot / 1e6), '%.1f' % (avg / 1e3), '%.1f' % (mn / 1e3), '%.1f' % (mx / 1e3),
                       '%.1f' % (100.0 * tot / total)] + regs + [lds]
        print('| ' + ' | '.join(str(v) for v in vals) + ' |')


if __name__ == '__main__':
    main()
