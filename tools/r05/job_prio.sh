#!/bin/bash
# wave-priority experiment on the feature-volume kernels: HEAD~ library vs the build with s_setprio (profiles/r05/experiments.md)
export TMPDIR=/tmp
A=$PWD/implicit-depth_amd/_obj/abl
for rep in 1 2; do
for cfg in "32 4 64" "32 7 64" "32 8 64" "16 12 64"; do
  echo "== $cfg head";  IDH_LIB=$A/libidh_fvhead.so python tools/perf_fv.py $cfg 10 2>&1 | grep -v amdgpu.ids
  echo "== $cfg prio";  python tools/perf_fv.py $cfg 10 2>&1 | grep -v amdgpu.ids
done
echo "== f16x3 head"; MLP_MATH=f16x3 IDH_LIB=$A/libidh_fvhead.so python tools/perf_fv.py 32 7 64 10 2>&1 | grep -v amdgpu.ids
echo "== f16x3 p1";   MLP_MATH=f16x3 IDH_LIB=$A/libidh_fv16p1.so python tools/perf_fv.py 32 7 64 10 2>&1 | grep -v amdgpu.ids
echo "== f16x3 p2";   MLP_MATH=f16x3 IDH_LIB=$A/libidh_fv16p2.so python tools/perf_fv.py 32 7 64 10 2>&1 | grep -v amdgpu.ids
done
