#!/bin/bash
# round-6 job 4: the MFMA-shape micro benchmark, the bench self-launch tests at the weak-scaling default, then the round's evidence run
O=gpurun_out/r06; mkdir -p $O
tools/micro/wino4_mfma_shape.bin > $O/job4_mfma_shape.txt 2>&1; cat $O/job4_mfma_shape.txt
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -k "bench" > $O/job4_bench_tests.log 2>&1; tail -3 $O/job4_bench_tests.log
bash tools/final_profiles.sh r06 > $O/job4_final_profiles.txt 2>&1; tail -60 $O/job4_final_profiles.txt
