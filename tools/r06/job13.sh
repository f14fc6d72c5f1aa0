#!/bin/bash
# round-6 job 13: F(4x4) V writes as 2 x ds_write_b128 + ds_write_b32 per k-step (HEAD) vs nine ds_write_b32 (-DIDH_W4_NO_VPACK); parity
O=gpurun_out/r06; mkdir -p $O
export LAYERS=0,1,2,3,5,6,8,10 VARIANTS=wino4
for rep in 1 2; do
  echo "== HEAD (rep $rep)"; python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
  echo "== NOVPACK (rep $rep)"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablw4_NOVPACK.so python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
done > $O/job13_w4_vpack.txt 2>&1; cat $O/job13_w4_vpack.txt
timeout 1500 python -m pytest tests/test_conv_wino4_gpu.py tests/test_operating_point_gpu.py tests/test_net_abi_gpu.py "tests/test_bdmodel_gpu.py::test_full_size_bdmodel_forward_golden" "tests/test_bdmodel_gpu.py::test_full_size_depthmodel_forward_golden" -q -m gpu > $O/job13_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/job13_tests.log
for rep in 1 2; do
timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 60 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench HEAD', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['worst_frame_vs_b1_rel'])"
IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablw4_NOVPACK.so timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 60 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench NOVPACK', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
