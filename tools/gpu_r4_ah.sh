#!/bin/bash
# round 4: what the memory system delivers for gathers of single doubles (tools/bench_micro/gather_bw.hip)
O=$GRAFT_REPO_ROOT/gpurun_out/r4ah; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 200 ./build/gather_bw > $O/gather.txt 2>&1 || { tail -5 $O/gather.txt; exit 1; }
cat $O/gather.txt
