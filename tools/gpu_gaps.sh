#!/bin/bash
# kernel durations and launch gaps of the train pipeline at a small lockstep group
O=$GRAFT_REPO_ROOT/gpurun_out/gaps
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 512 200 2 4 "$1" > $O/kt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kgaps.py $O/kt > $O/gaps_$1.txt; cat $O/kt.log | tail -2; cat $O/gaps_$1.txt
