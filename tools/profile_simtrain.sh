#!/bin/bash
# rocprofv3 kernel trace of the device-side simulator training step (tools/simtrain_rate.py) -> gpurun_out/<tag>_simtrain_kernel_stats.md; usage: profile_simtrain.sh <tag> [families]
tag=${1:-rXX}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_st -o st -- env ALGOS=${2:-dnn,widedeep,lstm,dien} python $repo/tools/simtrain_rate.py > $out/${tag}_simtrain.log 2>&1
db=$(find /tmp/prof_st -name '*.db' | head -1)
python $repo/tools/rocpd_summary.py $db > $out/${tag}_simtrain_kernel_stats.md 2>&1
python $repo/tools/rocpd_timeline.py $db k_emb_flatten 400 > $out/${tag}_simtrain_timeline.md 2>&1
