#!/bin/bash
# round 4: deterministic mode with ONE copy of the heavy sums (adds of a sum from one wave only): GPU suite, cost at 4096 and 10 000
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -s > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; grep -h "deterministic mode" $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; grep -n "^E " $O/tests.log | head; exit $rc; fi
for det in 0 1; do
  timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 deterministic=$det > $O/eval_det$det.txt 2>&1 || { tail -3 $O/eval_det$det.txt; exit 1; }
  echo "det $det: $(grep 'seq/s' $O/eval_det$det.txt | tail -1 | cut -c1-120)"
done
