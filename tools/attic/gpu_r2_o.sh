#!/bin/bash
# kernel durations and gaps of ONE engine evaluating a mini-batch (128 sequences), no loads in between
O=$GRAFT_REPO_ROOT/gpurun_out/r2o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 128 200 6 4 > $O/kt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kgaps.py $O/kt > $O/gaps.txt; tail -1 $O/kt.log; head -14 $O/gaps.txt
