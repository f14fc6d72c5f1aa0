#!/bin/bash
# position-split F(4x4) kernel: parity + A/B against conv3x3_wino4_k (IDH_W4_SPLIT=0)
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wino4_gpu.py -x -q 2>&1 | tail -8
export LAYERS=${LAYERS:-0,1,2,3,5,6,7} VARIANTS=wino4
for rep in 1 2; do
  echo "== split (rep $rep)"; timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
  echo "== old (rep $rep)"; IDH_W4_SPLIT=0 timeout 300 python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
done
