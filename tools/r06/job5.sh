#!/bin/bash
# round-6 job 5: cv_dot_win_k A/B (static unit copies, run-list pre-pass) + its tests; zero-volume pipeline test
O=gpurun_out/r06; mkdir -p $O
export IDH_LIB_ANY_ABI=1
for B in 32 8; do echo "=== B=$B"; bash tools/abl_dot2.sh run $B dyn nopre old; done 2>&1 | grep -v amdgpu.ids > $O/job5_dot_ab.txt; cat $O/job5_dot_ab.txt
python tools/perf_dot.py 3 32 7 96 2>&1 | grep kernel=; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_abldot_old.so python tools/perf_dot.py 3 32 7 96 2>&1 | grep kernel=
unset IDH_LIB_ANY_ABI
timeout 1500 python -m pytest tests/test_cost_volume_gpu.py tests/test_cost_volume_stress_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "not bench" > $O/job5_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/job5_tests.log
