#!/bin/bash
export TMPDIR=/tmp
OLD=$PWD/implicit-depth_amd/lib/ab/libidh_old.so
mkdir -p gpurun_out/c2
for b in 1 2 4; do
  for r in 1 2; do
    for l in old new; do
      if [ $l = old ]; then export IDH_LIB=$OLD; else unset IDH_LIB; fi
      IDH_LEVELS_ORDER=plan timeout 280 python tools/perf_levels.py $b > gpurun_out/c2/levels_b${b}_${l}.txt 2>&1
      echo "$l $(grep units= gpurun_out/c2/levels_b${b}_${l}.txt)"
    done
  done
done
unset IDH_LIB
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
