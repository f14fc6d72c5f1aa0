#!/bin/bash
# round 5, GPU job 3: zero-scratch F(4x4) kernel (three variants), residual prefetch, phase-offset experiment
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wino4_gpu.py tests/test_bdmodel_gpu.py tests/test_hot_path_split_gpu.py -x -q 2>&1 | tail -5 | tee $O/job3_pytest.txt
export LAYERS=${LAYERS:-0,1,2,3,5,8} VARIANTS=wino4
A=$PWD/implicit-depth_amd/_obj/abl
for rep in 1 2; do
  for lib in HEAD R04; do
    echo "== $lib (rep $rep)"
    if [ $lib = HEAD ]; then timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids; else IDH_LIB=$A/libidh_ablw4_$lib.so timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids; fi
  done
done | tee $O/job3_ab_head_vs_r04.txt
for d in 1000 2000 4000 6000; do echo "== delay $d cycles/stage"; IDH_W4_DELAY=$d timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids; done | tee $O/job3_delay.txt
for lib in NOEPI EPILIN; do
  echo "== $lib"; IDH_LIB=$A/libidh_ablw4_$lib.so timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
done | tee $O/job3_ablations.txt
timeout 600 python bench.py --no-cpu-baseline --no-split-line --no-extras 2>&1 | tail -1 | cut -c1-600 | tee $O/job3_bench.json
