#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/calib_fetch.hip): plain run for the timings, then one --pmc pass per counter.
O=$GRAFT_REPO_ROOT/gpurun_out/calib
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $GRAFT_REPO_ROOT/build/calib_fetch > $O/plain.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run -- $GRAFT_REPO_ROOT/build/calib_fetch > $O/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run -- $GRAFT_REPO_ROOT/build/calib_fetch > $O/write.log 2>&1 &&
python3 $GRAFT_REPO_ROOT/tools/calib_report.py $O > $O/report.txt 2>&1; cat $O/plain.log $O/report.txt
