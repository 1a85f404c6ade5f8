#!/usr/bin/env python
"""Derived table for a `<tag>_pmc.md` written by tools/profile_round.sh: matrix-pipe busy fraction per kernel and grid =
SQ_VALU_MFMA_BUSY_CYCLES / 32 (the counter as rocprofv3 reports it on this part divides by 32 to per-SIMD busy cycles: k_augru_x's
1.887e7 / 32 = 589.8 k = exactly its 18 432 MFMAs x 32 cycles, profiles/r05x_pmc.md) over the WALL cycles of the launch,
GRBM_GUI_ACTIVE - ONE denominator everywhere (SQ_BUSY_CYCLES is 6 - 8 % shorter than the wall and flattered the ratio;
it is kept in a column of its own).

    python tools/pmc_mfma_busy.py gpurun_out/<tag>_pmc.md      (prints markdown; profile_round.sh appends it to the file)"""
import sys


def main():
    rows = {}
    for line in open(sys.argv[1]):
        if not line.startswith('| `'):
            continue
        c = [x.strip() for x in line.strip().strip('|').split('|')]
        if len(c) < 7:
            continue
        key = (c[0], c[1], c[2])
        rows.setdefault(key, {})[c[3]] = (float(c[5]), float(c[6]), int(c[4]))
    print('## derived: matrix-pipe busy against wall cycles (SQ_VALU_MFMA_BUSY_CYCLES / 32 / GRBM_GUI_ACTIVE)')
    print('| kernel | grid_x | grid_y | MFMA busy / SIMD | wall cycles | busy fraction of the wall | (against SQ_BUSY_CYCLES) | avg_us |')
    print('|---|---|---|---|---|---|---|---|')
    out = []
    for (name, gx, gy), v in rows.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v and v['SQ_VALU_MFMA_BUSY_CYCLES'][0] > 0:
            busy = v['SQ_VALU_MFMA_BUSY_CYCLES'][0] / 32.0
            wall = v['GRBM_GUI_ACTIVE'][0]
            sq = v.get('SQ_BUSY_CYCLES', (0, 0, 0))[0]
            out.append((v['GRBM_GUI_ACTIVE'][1] * v['GRBM_GUI_ACTIVE'][2], name, gx, gy, busy, wall, busy / wall if wall else 0.0, busy / sq if sq else 0.0,
                        v['GRBM_GUI_ACTIVE'][1]))
    for _, name, gx, gy, busy, wall, f, fs, us in sorted(out, reverse=True)[:24]:
        print('| %s | %s | %s | %.4g | %.4g | **%.3f** | %.3f | %.1f |' % (name, gx, gy, busy, wall, f, fs, us))


if __name__ == '__main__':
    main()
