#!/bin/bash
# F(4x4) kernel at HEAD against the round-4 kernel (libidh_ablw4_R04.so, built from the r04 source), same box, interleaved
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wino4_gpu.py -x -q 2>&1 | tail -3
export LAYERS=${LAYERS:-0,1,2,3,5,6,7,8} VARIANTS=wino4
A=$PWD/implicit-depth_amd/_obj/abl
for rep in 1 2; do
  echo "== HEAD (rep $rep)"; timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
  echo "== R04 (rep $rep)"; IDH_LIB_ANY_ABI=1 IDH_LIB=$A/libidh_ablw4_R04.so timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
done
