#!/bin/bash
# round 4: sequences per lockstep group (does a group's working set between two diagonals fit the Infinity Cache?)
O=$GRAFT_REPO_ROOT/gpurun_out/r4am; mkdir -p $O
cd $GRAFT_REPO_ROOT
for g in 0 5000 2500 1250 625 400; do
  timeout -k 10 200 python tools/run_eval.py 10000 200 3 4 $g > $O/e.txt 2>&1 || { echo "failed: $g"; tail -3 $O/e.txt; exit 1; }
  echo "group $g: $(grep 'seq/s' $O/e.txt | tail -1 | cut -c1-100)"
done
