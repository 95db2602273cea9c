#!/bin/bash
# round 4: BPP filter from the candidate table: tests of the filter, load laps against the mask walk
O=$GRAFT_REPO_ROOT/gpurun_out/r4q; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "filter or bpp" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for walk in 0 1; do
  if [ $walk = 1 ]; then export ELEMDP_BPP_WALK=1; else unset ELEMDP_BPP_WALK; fi
  ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/laps_walk$walk.txt 2>&1 || { tail -5 $O/laps_walk$walk.txt; exit 1; }
  echo "walk $walk"; grep "BPP filter\|== load" $O/laps_walk$walk.txt | tail -8
done
unset ELEMDP_BPP_WALK
timeout -k 10 200 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan.txt 2>&1; tail -1 $O/scan.txt
