#!/bin/bash
# HBM traffic counters of one train evaluation (separate --pmc passes, no tracing domains).  Outputs under gpurun_out/pmc/.
O=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/fetch.log 2>&1 &&
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 2048 200 1 4 > $O/write.log 2>&1 &&
ls -R $O | head -20
