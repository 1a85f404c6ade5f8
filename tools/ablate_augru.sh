#!/bin/bash
# timing experiment: AUGRU kernel with parts removed (results are wrong for non-zero variants)
# needs a library built with the ablation variants: python -c "from rl4rs_amd.build import build_lib; build_lib(force=True, extra_flags=['-DRL4RS_ABLATE'])"
for v in 0 2 4 15; do
  RL4RS_AUGRU_ABLATE=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('ablate=$v  env-steps/s %.0f  augru avg_launch_ms %.3f  TF/s %.1f' % (d['value'], r['avg_launch_ms'], r['achieved']))"
done
