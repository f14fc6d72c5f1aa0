#!/bin/bash
# round-6 job 6: the whole GPU suite + smoke at HEAD, then the evidence run (tools/final_profiles.sh r06)
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/job6_all_tests.log 2>&1; echo "all tests rc=$?"; tail -3 $O/job6_all_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/job6_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/job6_smoke.log
bash tools/final_profiles.sh r06 > $O/job6_final_profiles.txt 2>&1; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp" $O/job6_final_profiles.txt | tail -45
