#!/bin/bash
# round 4: workgroups of several blocks (option nblk): GPU suite, then the evaluation of 4096 x L=200 for nblk = 1 .. 5
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
for nb in 1 2 3 4 5; do
  ELEMDP_LDS_DEBUG=1 timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 nblk=$nb > $O/eval_nblk$nb.txt 2>&1 || exit 1
  echo "nblk $nb: $(grep 'lin group' $O/eval_nblk$nb.txt | head -1)"; grep "seq/s" $O/eval_nblk$nb.txt | tail -1
done
timeout -k 10 200 python bench.py --steps 5 --warmup 2 > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-600
