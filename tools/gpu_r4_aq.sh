#!/bin/bash
# round 4: does the ORDER of the records inside a role segment matter (sorted by item index = grouped by outer cell: option sorted_plan)?
O=$GRAFT_REPO_ROOT/gpurun_out/r4aq; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for opt in "" "sorted_plan=1"; do
  timeout -k 10 200 python tools/run_eval.py 10000 200 3 4 0 $opt > $O/e.txt 2>&1 || { echo "failed: $opt"; tail -3 $O/e.txt; exit 1; }
  echo "[$opt] $(grep '^load' $O/e.txt) $(grep 'seq/s' $O/e.txt | tail -1 | cut -c1-110)"
done
done
