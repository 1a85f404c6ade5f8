#!/usr/bin/env python
"""Find loads the compiler serialised: per kernel, how many vector-memory loads are followed (within two instructions) by
``s_waitcnt vmcnt(0)`` - each one a full memory round trip nothing else is in flight behind - and how many spills / scratch
reloads sit in the kernel.  Two real finds of round 5 came from this scan: k_cat_attn2 (eight gathered values spilled behind
their loads because the sums that consume them had been sunk to the end of the kernel) and k_gemm_f32_t128's epilogue (one
addend load, wait, activation and store per element).

    python tools/isa_wait_scan.py [unit.hip ...]        (default: every unit of rl4rs_amd/build.py; needs hipcc, no GPU)

The ISA comes from ``hipcc -S --cuda-device-only`` with the flags of the in-tree build."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4rs_amd.build import CSRC, SOURCES, NO_SLP  # noqa: E402

LOADS = ('global_load', 'buffer_load', 'scratch_load', 'flat_load')


def scan(path):
    """serial: loads with a vmcnt(0) wait within two instructions; loops: short inner loops (<= 48 instructions) that hold a load and
    a vmcnt(0) wait - a staging loop the compiler left as one memory round trip per iteration.  polls: the same two patterns when
    every load involved is an agent-scope atomic load (``sc1``) - a spin on a flag another workgroup writes (the grid barrier of
    k_ppo_pass) IS a round trip per look by construction; counted apart so that they do not hide a real find."""
    stats, name, window = {}, None, []
    loop_start = None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, window, loop_start = m.group(1), [], None
            stats[name] = dict(serial=0, loads=0, waits0=0, spills=0, loops=0, polls=0)
            continue
        if name is not None and 'Inner Loop Header' in line:
            loop_start = len(window)
        t = line.strip()
        if t.startswith('.Lfunc_end'):
            name = None
        if name is None or not t or t[0] in ';.':
            continue
        st = stats[name]
        if t.startswith(LOADS):
            st['loads'] += 1
        if 'Folded Spill' in t:
            st['spills'] += 1
        if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
            st['waits0'] += 1
            near = [x for x in window[-2:] if x.startswith(LOADS)]
            if near:
                st['polls' if all(' sc1' in x for x in near) else 'serial'] += 1
        window.append(t)
        if loop_start is not None and t.startswith('s_cbranch'):
            body = window[loop_start:]
            lds = [x for x in body if x.startswith(LOADS)]
            if len(body) <= 48 and lds and any(x.startswith('s_waitcnt') and 'vmcnt(0)' in x for x in body):
                if not all(' sc1' in x for x in lds):
                    st['loops'] += 1
            loop_start = None
    return stats


def main():
    units = sys.argv[1:] or SOURCES
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for u in units:
            out = os.path.join(d, os.path.basename(u).replace('.hip', '.s'))
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-mllvm', '-pragma-unroll-threshold=200000',
                   '-S', '--cuda-device-only', '-I', os.path.join(os.path.dirname(CSRC), '..', 'include'), '-o', out,
                   os.path.join(CSRC, os.path.basename(u))] + (['-fno-slp-vectorize'] if os.path.basename(u) in NO_SLP else [])
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for k, st in scan(out).items():
                if st['serial'] >= 3 or st['spills'] or st['loops'] or st['polls']:
                    rows.append((st['serial'], st['spills'], st['loads'], st['waits0'], k, os.path.basename(u), st['loops'], st['polls']))
    demangle = subprocess.run(['c++filt'] + [r[4] for r in rows], capture_output=True, text=True).stdout.split('\n') if rows else []
    print('| serialised loads | one-round-trip-per-iteration loops | vector spills | spin polls | loads | vmcnt(0) waits | kernel | unit |\n|---|---|---|---|---|---|---|---|')
    for r, dn in sorted(zip(rows, demangle), reverse=True):
        print('| %d | %d | %d | %d | %d | %d | `%s` | %s |' % (r[0], r[6], r[1], r[7], r[2], r[3], dn[:100], r[5]))


if __name__ == '__main__':
    main()
