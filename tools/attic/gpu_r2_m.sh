#!/bin/bash
# scan: parity subset + timing at the config-E shape
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_round2_gpu.py -q -m gpu -x > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2m/pytest.log
timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" 2>&1 | tail -1
timeout -k 10 300 python tools/scan_bench.py 10000 200 2>&1 | tail -1
