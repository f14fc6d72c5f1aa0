#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_net_abi_gpu.py tests/test_mlp_gpu.py tests/test_pipeline_gpu.py "tests/test_bdmodel_gpu.py::test_bench_call_shape_workloads_match_reference_bdmodel" -x -q -m gpu -s > $O/job2_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/job2_new_tests.log
tail -5 $O/job2_new_tests.log
timeout 1200 python bench.py > $O/job2_bench.json 2> $O/job2_bench.err; echo "bench rc=$? lines=$(wc -l < $O/job2_bench.json)"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/job2_bench.json').readline())
print("value", d["value"], "extra", json.dumps(d.get("extra"))[:1500])
PY
