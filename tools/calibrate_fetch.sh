#!/bin/bash
# FETCH_SIZE calibration for the AUGRU kernel's own access pattern (MI355X_MICROARCH.md: "calibrate on a known byte count in
# your own access pattern"): tools/augru_bench.py runs the recurrence with 4096 DISTINCT cache slots per sequence input, so an
# obs-sized launch must read every projection row exactly once: 8192 row-inputs x 64 steps x 768 floats = 1 610.6 MB
# (+ 1.6 MB of weights, 2 MB of attention scores, 16.8 MB of first-GRU-independent state: none).  The ratio
# raw FETCH_SIZE / 1 610.6 MB is the correction factor for this kernel's LDS-DMA (16 B per lane) stream.
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_cal -o cal -- python $repo/tools/augru_bench.py x --reps 2 > /tmp/cal.log 2>&1
db=$(find /tmp/prof_cal -name '*.db' | head -1)
python $repo/tools/rocpd_pmc.py $db 2>&1 | grep -i "augru\|kernel \|---" | head -8
