#!/bin/bash
# round 4: role lists built per (sequence, role) workgroup with LDS counters: full GPU suite, load laps, streamed train
O=$GRAFT_REPO_ROOT/gpurun_out/r4s; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for g in 0 1; do
  if [ $g = 1 ]; then export ELEMDP_ROLE_GLOBAL=1; else unset ELEMDP_ROLE_GLOBAL; fi
  ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/laps_g$g.txt 2>&1 || { tail -5 $O/laps_g$g.txt; exit 1; }
  echo "role passes with global atomics: $g"; grep "plan\|== load" $O/laps_g$g.txt | tail -6
done
unset ELEMDP_ROLE_GLOBAL
timeout -k 10 300 python tools/stream_60k.py 60000 200 10000 > $O/stream.txt 2>&1; tail -4 $O/stream.txt
