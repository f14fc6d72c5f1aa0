#!/bin/bash
# (record of a tuning build: IDH_SPLIT_W3 was read by csrc/conv.hip only while this sweep ran; the shipped kernel has the weight as a constant)
export TMPDIR=/tmp
for b in 1 4; do
 for w in 3 4 6 9; do for pw in 1.0 0.5 0.25; do
   echo "B=$b W3=$w PROJ_W=$pw $(IDH_SPLIT_W3=$w IDH_PROJ_W=$pw timeout 280 python tools/perf_levels.py $b 2>&1 | grep units=)"
 done; done
done
