#!/bin/bash
# round 3, step a: GPU parity suite on the compact tables, then a quick rate check
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r3a/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 gpurun_out/r3a/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/run_eval.py 4096 200 3 4 > gpurun_out/r3a/eval4096.log 2>&1; echo "eval rc $?"; tail -4 gpurun_out/r3a/eval4096.log
