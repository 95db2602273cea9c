#!/bin/bash
# round 4: BPP filter with one workgroup per sequence: tests of the filter, load laps against the diagonal launches
O=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "filter or bpp" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for mode in seq diag; do
  if [ $mode = diag ]; then export ELEMDP_BPP_DIAG=1; else unset ELEMDP_BPP_DIAG; fi
  ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/laps_$mode.txt 2>&1 || { tail -5 $O/laps_$mode.txt; exit 1; }
  echo "mode $mode"; grep "BPP filter\|== load" $O/laps_$mode.txt | tail -4
done
unset ELEMDP_BPP_DIAG
timeout -k 10 200 python tools/minibatch_bench.py 2000 200 200 > $O/mb.txt 2>&1; tail -1 $O/mb.txt
