#!/bin/bash
# quick: parity subset, eval timing at 4096 and 1250 and 128 sequences
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_round2_gpu.py -q -m gpu -x > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2f/pytest.log
timeout -k 10 200 python tools/run_eval.py 4096 200 3 4 2>&1 | tail -1
timeout -k 10 200 python tools/run_eval.py 1250 200 3 4 2>&1 | tail -1
timeout -k 10 200 python tools/run_eval.py 128 200 4 4 2>&1 | tail -1
