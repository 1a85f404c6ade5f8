#!/bin/bash
# Same-box A/B of scorer kernel options on the default bench workload: for each RL4RS_DIEN_OPTS value given (use "-" for the
# defaults) run bench.py twice, alternating, and print ms_per_step + the per-kernel breakdown.
# usage: tools/ab_opts.sh - cat_v1 [...]      -> gpurun_out/ab_opts.txt
out=gpurun_out/ab_opts.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2; do
  for o in "$@"; do
    if [ "$o" = "-" ]; then opts=""; else opts="$o"; fi
    RL4RS_DIEN_OPTS="$opts" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r['kernels']
print('opts=%-12s rep=$rep ms_per_step=%.3f  ' % ('$o', r['ms_per_step']) + '  '.join('%s=%.3f' % (n.split('(')[0][:14], v['ms']) for n,v in k.items()))
" | tee -a $out
  done
done
