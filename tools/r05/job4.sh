#!/bin/bash
# round 5, GPU job 4: fv_mlp_k with the mask columns folded into the bias (-24 MFMAs per plane at K = 7) and LeakyReLU as mul + max
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_feature_volume_gpu.py tests/test_bdmodel_gpu.py tests/test_mlp_split_gpu.py tests/test_hot_path_head_gpu.py -x -q 2>&1 | tail -5 | tee $O/job4_pytest.txt
for i in 1 2; do timeout 300 python tools/perf_fv.py 32 7 64 10 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/perf_fv.py 32 8 64 10 2>&1 | grep -v amdgpu.ids; done | tee $O/job4_fv.txt
timeout 600 python bench.py --no-cpu-baseline --no-split-line --no-extras 2>&1 | tail -1 | cut -c1-700 | tee $O/job4_bench.json
