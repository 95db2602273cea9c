#!/bin/bash
# kernel durations and gaps of the mini-batch training mode (64 + 64 sequences per iteration)
O=$GRAFT_REPO_ROOT/gpurun_out/mb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/minibatch_bench.py 2000 200 10 > $O/kt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kgaps.py $O/kt > $O/gaps.txt; tail -1 $O/kt.log; head -30 $O/gaps.txt
