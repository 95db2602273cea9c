#!/bin/bash
# round 4: laps of a streamed scan (50 000 x L=300 in chunks of 12 500), second (warm) pass
O=$GRAFT_REPO_ROOT/gpurun_out/r4ad; mkdir -p $O
cd $GRAFT_REPO_ROOT
ELEMDP_TIME=1 timeout -k 10 300 python tools/stream_scan2.py 50000 300 12500 > $O/laps.txt 2>&1 || { tail -5 $O/laps.txt; exit 1; }
sed -n '/==== pass 1/,$p' $O/laps.txt | grep -v "load: \(weights\|host\)" | tail -50; grep "^pass" $O/laps.txt
