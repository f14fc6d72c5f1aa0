#!/bin/bash
export TMPDIR=/tmp
for b in 1 4; do
 for mc in 3 4 6 8; do for mx in 16; do
   echo "B=$b SPLIT_MIN_CHUNKS=$mc SPLIT_MAX=$mx $(timeout 280 python tools/perf_levels.py $b 400 0 1 $mc $mx 2>&1 | grep units=)"
 done; done
done
