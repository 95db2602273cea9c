#!/bin/bash
# round 4: records fetched ahead of the pair phases (no barrier / round trip of their own): GPU suite, evaluation of 4096 x L=200
# for one and three blocks per workgroup, phase ceilings and in-kernel phase clocks of both, mini-batch iteration, scan
O=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
for nb in 1 3; do
  ELEMDP_NBLK=$nb ELEMDP_LDS_DEBUG=1 timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 > $O/eval_nblk$nb.txt 2>&1 || exit 1
  echo "nblk $nb: $(grep 'lin group' $O/eval_nblk$nb.txt | head -1 | sed 's/.*lds/lds/')"; grep "seq/s" $O/eval_nblk$nb.txt | tail -1
  ELEMDP_NBLK=$nb timeout -k 10 200 python tools/dbg_phases.py 4096 200 0 1 2 4 7 2055 1031 > $O/ceil_nblk$nb.txt 2>&1; cat $O/ceil_nblk$nb.txt
  ELEMDP_NBLK=$nb timeout -k 10 200 python tools/prof_phases.py 4096 200 > $O/phases_nblk$nb.txt 2>&1; cat $O/phases_nblk$nb.txt
done
timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan.txt 2>&1; tail -4 $O/scan.txt
timeout -k 10 300 python tools/minibatch_bench.py > $O/mb.txt 2>&1; tail -3 $O/mb.txt
