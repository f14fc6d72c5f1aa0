#!/bin/bash
# round-6 job 1: new tests first, then the whole GPU suite, then the default bench line
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_operating_point_gpu.py tests/test_pipeline_gpu.py tests/test_conv_gpu.py tests/test_mlp_gpu.py -x -q -m gpu -s > $O/job1_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/job1_new_tests.log
tail -5 $O/job1_new_tests.log
timeout 1200 python bench.py > $O/job1_bench.json 2> $O/job1_bench.err; echo "bench rc=$?"
cut -c1-1500 $O/job1_bench.json
timeout 1500 python -m pytest tests -q -m gpu -x > $O/job1_all_tests.log 2>&1; echo "all tests rc=$?"; tail -3 $O/job1_all_tests.log
