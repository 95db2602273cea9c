#!/bin/bash
# full GPU suite, then the evidence run (bench line, kernel trace, traffic counters)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r2p/pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r2p/pytest.log
bash tools/gpu_r2_profile.sh
