#!/bin/bash
# Same-box A/B of library variants built with tools/build_ab.py: tools/ab_bench.sh <rounds> head <variant> <variant> ...
# prints env-steps/s, ms per episode-batch, AUGRU ms per launch (timed region) and per episode-batch (breakdown pass)
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = head ]; then L=""; else L="tools/_ab/$v/librl4rs_hip.so"; fi
    RL4RS_LIB=$L timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-14s %9d  %.3f ms  augru %.4f / %.3f' % ('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernels']['k_augru_x']['ms']))"
  done
done
