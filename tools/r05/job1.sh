#!/bin/bash
# round 5, GPU job 1: parity of the re-allocated F(4x4) kernel + ablation bounds for the review's items 1(a)-(c)
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wino4_gpu.py -x -q 2>&1 | tail -5 | tee $O/job1_pytest.txt
export LAYERS=${LAYERS:-0,1,2,3,5} VARIANTS=wino4
A=$PWD/implicit-depth_amd/_obj/abl
for rep in 1 2; do
  for lib in HEAD R04; do
    echo "== $lib (rep $rep)"
    if [ $lib = HEAD ]; then timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids; else IDH_LIB=$A/libidh_ablw4_$lib.so timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids; fi
  done
done | tee $O/job1_ab_head_vs_r04.txt
for lib in HALFA EPILIN HALFA_EPILIN NOXFORM NOEPI NOA NOHALO; do
  echo "== $lib"; IDH_LIB=$A/libidh_ablw4_$lib.so timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
done | tee $O/job1_ablations.txt
timeout 300 python tools/perf_fv.py 32 7 64 10 2>&1 | grep -v amdgpu.ids | tee $O/job1_fv.txt
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -2 | tee $O/job1_bench.json
