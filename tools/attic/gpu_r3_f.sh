#!/bin/bash
# kernel durations and launch gaps of a mini-batch-sized evaluation (128 sequences; one stream so that gaps are meaningful)
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 128 200 4 4 0 group_streams=1 > $O/kt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kgaps.py $O/kt > $O/gaps.txt; tail -2 $O/kt.log; head -14 $O/gaps.txt
rm -rf $O/kt
