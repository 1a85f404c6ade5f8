#!/bin/bash
# rocprofv3 kernel trace only of the default bench workload -> gpurun_out/<tag>_kernel_stats.md (usage: profile_kt.sh <tag> [RL4RS_DIEN_OPTS])
tag=${1:-rXX}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt_$tag
RL4RS_DIEN_OPTS="$2" timeout -k 10 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$tag -o kt -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra-legs > $out/${tag}_kt.log 2>&1
db=$(find /tmp/prof_kt_$tag -name '*.db' | head -1)
python $repo/tools/rocpd_summary.py $db > $out/${tag}_kernel_stats.md 2>&1
