#!/bin/bash
# Round-end measurement on the GPU box: kernel trace + separate PMC passes of the default bench workload.
# usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*.md
tag=${1:-rXX}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra-legs"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $cmd > $out/${tag}_kt.log 2>&1
db=$(find /tmp/prof_kt -name '*.db' | head -1)
python $repo/tools/rocpd_summary.py $db > $out/${tag}_kernel_stats.md 2>&1
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/prof_pmc$i -o pmc -- $cmd > $out/${tag}_pmc$i.log 2>&1
  db=$(find /tmp/prof_pmc$i -name '*.db' | head -1)
  echo "## pass: $pmc" >> $out/${tag}_pmc.md
  python $repo/tools/rocpd_pmc.py $db 2>&1 | head -40 >> $out/${tag}_pmc.md
  echo >> $out/${tag}_pmc.md
done
python $repo/tools/pmc_mfma_busy.py $out/${tag}_pmc.md >> $out/${tag}_pmc.md 2>&1
