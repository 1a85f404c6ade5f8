"""Build librl4rs_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'librl4rs_hip.so')
SOURCES = ['env.hip', 'gemm.hip', 'dien.hip', 'augru_x.hip', 'policy.hip', 'records.hip', 'step.hip']
# Units whose MFMA kernels run VALU epilogues beside another wave's MFMAs are compiled without SLP vectorisation: packed fp32 VALU
# (v_pk_fma / add / mul_f32) serialises with the matrix pipe (tools/mfma_valu_overlap.hip, profiles/r04p_mfma_valu_overlap.txt;
# same-box A/B: k_cat_attn2 -8 %, k_din_x -2 %, end to end +0.7 %).  The other units keep it (the learners' element-wise and
# reduction kernels are 1 - 4 % faster with it) - and so does augru_x.hip: k_augru_x<2,4,2> sits at the register limit and spills without it.
NO_SLP = ('dien.hip', 'gemm.hip')
HEADERS = ['common.hpp', 'recur_args.hpp', 'augru_x.hpp', 'din_x.hpp', 'recur_train.hpp', 'recur8.hpp', 'mfma4.hpp', 'simnet.hpp', 'gather_kernels.hpp', 'simtrain.hpp', 'dientrain.hpp', 'rawtrain.hpp', 'qlearn.hpp', 'contirl.hpp', 'amlp_fused.hpp', 'ppo_pass.hpp', 'policy_tile_std.hpp', os.path.join('..', '..', 'include', 'rl4rs_hip.h')]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source into one shared object. Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-ffp-contract=off',
               '-mllvm', '-pragma-unroll-threshold=200000',     # k_augru_h16: 48 weight items per step, fully unrolled
               '-c', os.path.join(CSRC, src), '-o', obj] + (['-fno-slp-vectorize'] if src in NO_SLP else []) + list(extra_flags)
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose=True))
