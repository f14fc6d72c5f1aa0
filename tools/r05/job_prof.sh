#!/bin/bash
# rocprofv3 kernel stats of the default bench step (hot path, B = 32) -> gpurun_out/r05/prof_hot
export TMPDIR=/tmp
ROOT=$PWD; O=$ROOT/gpurun_out/r05; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hot -o hot -- python $ROOT/bench.py --no-cpu-baseline --no-split-line --no-extras --steps 30 --warmup 3 > $O/prof_hot.log 2>&1 )
f=$(find $O/prof_hot -name "*kernel_stats.csv" | head -1); head -40 "$f" | cut -c1-220
tail -1 $O/prof_hot.log | cut -c1-400
# keep the merged output small: the raw trace is not needed
find $O/prof_hot -name "*kernel_trace.csv" -size +20M -delete
