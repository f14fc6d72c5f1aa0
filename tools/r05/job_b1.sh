#!/bin/bash
# one-frame plan: per-level table in plan order, kernel trace of the temporal loop, and the default bench line at HEAD
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/b1; mkdir -p $OUT
cd $ROOT
IDH_LEVELS_ORDER=plan timeout 300 python tools/perf_levels.py 1 > $OUT/levels_b1.txt 2>&1
IDH_LEVELS_ORDER=plan timeout 300 python tools/perf_levels.py 4 > $OUT/levels_b4.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-split-line --no-extras --workload temporal --steps 64 | tail -1 > $OUT/temporal.json
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
head -3 $OUT/levels_b1.txt; cut -c1-300 $OUT/temporal.json; cut -c1-400 $OUT/bench_default.json
