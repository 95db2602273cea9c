#!/bin/bash
# round 4: the full bench line after the load work (filter per sequence, plan passes, pinned staging)
O=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cat $O/bench.json
