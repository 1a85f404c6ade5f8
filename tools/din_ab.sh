#!/bin/bash
# k_din_x timing ablations (tools/build_ab.py dinab<k> -DRL4RS_DINX_AB=<k>): DIN ms per episode-batch from the bench breakdown
for v in head "$@"; do
  if [ $v = head ]; then L=""; else L=tools/_ab/$v/librl4rs_hip.so; fi
  RL4RS_LIB=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['kernels']
print('%-10s' % '$v', '  '.join('%s=%.3f' % (n[:12], v['ms']) for n,v in k.items() if 'din' in n or 'cat' in n or 'gru_h16' in n))"
done
