#!/bin/bash
# rocprofv3 kernel trace of the on-device training loops (bench.py --train a2c|ppo): gpurun_out/<tag>_train_<algo>_kernel_stats.md
# (round 1: `rocprofv3 --kernel-trace --stats` around --train a2c did not return within 15 minutes; every call is bounded here)
tag=${1:-rXX}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for algo in a2c ppo; do
  rm -rf /tmp/prof_tr_$algo
  cmd="python $repo/bench.py --train $algo --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-fp32-leg"
  timeout -k 10 150 rocprofv3 --kernel-trace -d /tmp/prof_tr_$algo -o tr -- $cmd > $out/${tag}_train_${algo}.log 2>&1
  echo "rc=$?" >> $out/${tag}_train_${algo}.log
  db=$(find /tmp/prof_tr_$algo -name '*.db' | head -1)
  if [ -n "$db" ]; then python $repo/tools/rocpd_summary.py $db > $out/${tag}_train_${algo}_kernel_stats.md 2>&1; fi
done
