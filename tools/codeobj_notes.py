#!/usr/bin/env python
"""Register / LDS / spill counts of every kernel in a HIP shared object, from the code-object metadata itself
(`llvm-readelf --notes` of the gfx950 ELFs inside `.hip_fatbin`): the numbers an occupancy argument has to rest on.
rocprofv3's `vgpr_count` column reports the ALLOCATION GRANULE-rounded arch-VGPR half of a unified register file
(120 / 128 for k_augru_x where the code object says 234 / 254 - VERDICT r3 weak #5).

usage: python tools/codeobj_notes.py [rl4rs_amd/csrc/librl4rs_hip.so] [--md]       (importable: kernel_table(path))
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def _gfx950_elfs(so_path):
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so_path])
        blob = open(fat, 'rb').read()
    out = []
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    pos = 0
    while True:
        base = blob.find(magic, pos)
        if base < 0:
            break
        n = struct.unpack_from('<Q', blob, base + len(magic))[0]
        p = base + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx950' in triple and size:
                out.append(blob[base + off:base + off + size])
        pos = base + len(magic)
    return out


def kernel_table(so_path):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, lds_static, scratch, vgpr_spill, sgpr_spill, wavefront)}"""
    table = {}
    for elf in _gfx950_elfs(so_path):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(elf)
            f.flush()
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if k == 'agpr_count':
                cur = {'agpr': int(v)}                      # first key of a kernel's block (keys are sorted)
            elif cur is not None:
                if k == 'group_segment_fixed_size': cur['lds_static'] = int(v)
                elif k == 'name': cur['mangled'] = v
                elif k == 'private_segment_fixed_size': cur['scratch'] = int(v)
                elif k == 'sgpr_count': cur['sgpr'] = int(v)
                elif k == 'sgpr_spill_count': cur['sgpr_spill'] = int(v)
                elif k == 'vgpr_count': cur['vgpr'] = int(v)
                elif k == 'vgpr_spill_count': cur['vgpr_spill'] = int(v)
                elif k == 'wavefront_size':
                    cur['wavefront'] = int(v)
                    if 'mangled' in cur:
                        table[cur['mangled']] = cur
                    cur = None
    if not table:
        return {}
    names = list(table)
    dem = subprocess.run(['c++filt'], input='\n'.join(names) + '\n', capture_output=True, text=True).stdout.splitlines()
    return dict((d, table[m]) for m, d in zip(names, dem))


def lookup(table, rocprof_name):
    """code-object entry of a kernel as rocprofv3 names it (demangled, sometimes with a leading 'void ')"""
    n = rocprof_name[5:] if rocprof_name.startswith('void ') else rocprof_name
    if n in table:
        return table[n]
    base = n.split('(')[0]
    hits = [v for k, v in table.items() if k.split('(')[0] == base or k.split('(')[0] == 'void ' + base]
    return hits[0] if len(hits) == 1 else None


def main():
    so = [a for a in sys.argv[1:] if not a.startswith('--')]
    so = so[0] if so else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rl4rs_amd', 'csrc', 'librl4rs_hip.so')
    t = kernel_table(so)
    print('| kernel | vgpr | agpr | sgpr | static LDS B | scratch B | vgpr spills | sgpr spills |')
    print('|---|---|---|---|---|---|---|---|')
    for name in sorted(t, key=lambda k: -t[k].get('vgpr', 0)):
        e = t[name]
        short = name if len(name) <= 100 else name[:97] + '...'
        print('| `%s` | %d | %d | %d | %d | %d | %d | %d |' % (short, e.get('vgpr', 0), e.get('agpr', 0), e.get('sgpr', 0), e.get('lds_static', 0),
                                                             e.get('scratch', 0), e.get('vgpr_spill', 0), e.get('sgpr_spill', 0)))


if __name__ == '__main__':
    main()
