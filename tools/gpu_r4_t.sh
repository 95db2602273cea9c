#!/bin/bash
# round 4: laps of the streamed train evaluation (60 000 x L=200, chunks of 10 000), resident reference, mini-batch mode
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
cd $GRAFT_REPO_ROOT
ELEMDP_TIME=1 STREAM_REPS=3 timeout -k 10 300 python tools/stream_60k.py 60000 200 10000 2>&1 | python tools/stream_laps.py > $O/stream_laps.txt 2>&1; cat $O/stream_laps.txt
timeout -k 10 300 python tools/minibatch_bench.py 2000 200 200 > $O/mb.txt 2>&1; tail -5 $O/mb.txt
