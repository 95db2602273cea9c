#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r2h/pytest.log
ELEMDP_TIME=1 timeout -k 10 300 python tools/load_laps.py 10000 300 2>&1 | grep -v "unfiltered" | tail -14
ELEMDP_TIME=1 timeout -k 10 300 python tools/load_laps.py 128 200 "((.*.))" 2>&1 | tail -8
timeout -k 10 300 python tools/minibatch_bench.py 2000 200 30 2>&1 | tail -3
