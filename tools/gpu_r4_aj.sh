#!/bin/bash
# round 4: padding of the compact tables' rows (option row_pad; 8 = rows start on 64-byte lines): timing at 4096 and 10 000
O=$GRAFT_REPO_ROOT/gpurun_out/r4aj; mkdir -p $O
cd $GRAFT_REPO_ROOT
for n in 4096 10000; do
for rep in 1 2; do
for opt in "row_pad=8" "row_pad=4" "row_pad=2" "row_pad=1"; do
  timeout -k 10 200 python tools/run_eval.py $n 200 3 4 0 $opt > $O/e.txt 2>&1 || { echo "failed: $opt"; tail -3 $O/e.txt; exit 1; }
  echo "n=$n [$opt] $(grep 'seq/s' $O/e.txt | tail -1 | cut -c1-130)"
done
done
done
