#!/usr/bin/env python
"""Launch-by-launch timeline of the LAST repetition of a workload in a rocprofv3 rocpd db: kernels in start order with their
duration and the idle gap in front of each - where a chain of small dependent launches spends its time (the tail of the simulator
training step: ~90 launches per step, tools/profile_simtrain.sh).

    python tools/rocpd_timeline.py <results.db> <anchor kernel substring> [max rows]

The window printed runs from the second-to-last to the last launch of the first kernel whose name contains the anchor (e.g. the
first kernel of a step)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2]
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
    st = 'start' if 'start' in cols else 'start_timestamp'
    en = 'end' if 'end' in cols else 'end_timestamp'
    rows = db.execute("select name, %s, %s, grid_x, grid_y, workgroup_x from kernels order by %s" % (st, en, st)).fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 2:
        print('anchor %r found %d times' % (anchor, len(idx)))
        return
    # the anchor may be launched several times per repetition: take the last two launches that are at least 100 kernels apart
    hi = idx[-1]
    lo = max(i for i in idx if hi - i >= 20)
    win = rows[lo:hi]
    t0 = win[0][1]
    busy = sum(r[2] - r[1] for r in win)
    span = win[-1][2] - t0
    print('window: %d launches, span %.3f ms, kernel time %.3f ms, idle %.3f ms' % (len(win), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    print('| # | at us | gap us | dur us | grid | kernel |\n|---|---|---|---|---|---|')
    prev_end = t0
    for i, (name, s, e, gx, gy, wx) in enumerate(win[:cap]):
        print('| %d | %.1f | %.1f | %.1f | %dx%d/%d | `%s` |' % (i, (s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, gx // max(wx, 1), gy, wx, name[:80]))
        prev_end = max(prev_end, e)


if __name__ == '__main__':
    main()
