#!/bin/bash
# round 3, step i: full GPU suite after the plan-builder change, then load / scan / mini-batch timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
ELEMDP_TIME=1 timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan.log 2>&1; echo "scan rc $?"; grep "plan of the filtered\|BPP filter, linear\|host arrays" $O/scan.log | tail -5; tail -1 $O/scan.log
timeout -k 10 300 python tools/minibatch_bench.py 2000 200 40 2>&1 | tail -1
timeout -k 10 300 python tools/run_eval.py 4096 200 3 4 2>&1 | tail -2
