"""Build a VARIANT of librl4rs_hip.so into tools/_ab/<name>/ (git-ignored, travels with gpurun) for same-box A/B runs:
    python tools/build_ab.py <name> [-DFLAG ...]
then on the GPU box: RL4RS_LIB=tools/_ab/<name>/librl4rs_hip.so python bench.py ...
AB_NO_SLP=augru_x.hip,... adds units to the ones compiled with -fno-slp-vectorize (AB_SLP=...: removes)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4rs_amd.build import CSRC, SOURCES, NO_SLP  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ab', name)
    os.makedirs(out, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    no_slp = (set(NO_SLP) | set(filter(None, os.environ.get('AB_NO_SLP', '').split(',')))) - set(filter(None, os.environ.get('AB_SLP', '').split(',')))
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(out, src.replace('.hip', '.o'))
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-ffp-contract=off',
               '-mllvm', '-pragma-unroll-threshold=200000', '-c', os.path.join(CSRC, src), '-o', obj] + (['-fno-slp-vectorize'] if src in no_slp else []) + flags
        procs.append(subprocess.Popen(cmd))
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit('hipcc failed')
    lib = os.path.join(out, 'librl4rs_hip.so')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    for o in objs:
        os.remove(o)
    print(lib)


if __name__ == '__main__':
    main()
