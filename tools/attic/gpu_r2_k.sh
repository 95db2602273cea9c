#!/bin/bash
# BPP filter: parity subset, load laps at the scan and the train shapes, mini-batch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_round2_gpu.py -q -m gpu -x > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2k/pytest.log
ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 10000 300 2>&1 | grep -E "load_batch|lap" | tail -12
ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 128 200 "((.*.))" 2>&1 | grep -E "load_batch|lap" | tail -8
