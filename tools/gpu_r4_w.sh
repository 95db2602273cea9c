#!/bin/bash
# round 4: plan builder (copies written by the role pass, popcount count pass), table-driven position weights: full GPU suite, load laps, streamed train
O=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/laps.txt 2>&1 || { tail -5 $O/laps.txt; exit 1; }
grep "elemdp\|== load" $O/laps.txt | tail -9
timeout -k 10 300 python tools/stream_60k.py 60000 200 10000 > $O/stream.txt 2>&1; tail -3 $O/stream.txt
timeout -k 10 200 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan.txt 2>&1; tail -1 $O/scan.txt
