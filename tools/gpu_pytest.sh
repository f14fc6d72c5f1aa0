#!/bin/bash
# quick GPU parity run: tools/gpu_pytest.sh [pytest args]
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -30 | tee gpurun_out/pytest_last.txt
