#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprofv3 kernel stats.
# Usage: tools/gpu_check.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo ==" ; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|gfx" | head -4
echo "== pytest -m gpu ==" 
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== smoke ==" 
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench ==" 
timeout 600 python bench.py 2>&1 | tail -3 | tee $OUT/bench.json
echo "== rocprofv3 kernel stats ==" 
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --no-cpu-baseline --steps 30 > $OLDPWD/$OUT/prof_bench.log 2>&1 )
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | tee $OUT/kernel_stats_head.csv
ls $OUT/prof/* | head
