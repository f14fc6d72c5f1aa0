#!/bin/bash
# round 5, GPU job 2: the whole GPU suite with per-test durations (the suite has to stay well under the driver's 1200 s)
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=60 2>&1 | tail -90 | tee $O/job2_pytest_durations.txt
