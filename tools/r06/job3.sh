#!/bin/bash
# round-6 job 3: where a small-batch F(2x2) launch goes (trace), the per-level table at B = 4 / 1, cv_dot_win_k PMC + trace at HEAD
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_net_abi_gpu.py tests/test_mlp_gpu.py tests/test_pipeline_gpu.py "tests/test_bdmodel_gpu.py::test_bench_call_shape_workloads_match_reference_bdmodel" -q -m gpu -s > $O/job3_new_tests.log 2>&1; echo "new tests rc=$?"; tail -4 $O/job3_new_tests.log
export IDH_LIB_ANY_ABI=1
for cfg in "64 64 96 128 2 4" "64 64 192 256 2 1" "128 128 48 64 2 4" "64 64 96 128 2 32"; do
  bash tools/trace_wino.sh run $cfg >> $O/job3_trace_wino.txt 2>&1
done
tail -30 $O/job3_trace_wino.txt
python tools/perf_levels.py 4 > $O/job3_levels_b4.txt 2>&1; head -50 $O/job3_levels_b4.txt
python tools/perf_levels.py 1 > $O/job3_levels_b1.txt 2>&1; head -3 $O/job3_levels_b1.txt
python tools/perf_dot.py 3 32 > $O/job3_perf_dot_b32.txt 2>&1; tail -5 $O/job3_perf_dot_b32.txt
