#!/bin/bash
# round 4: one-block kernels as kernels of their own (template MB), scan at one block per workgroup; k5_cyk cells-per-workgroup sweep
O=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
for nb in 0 1; do
  ELEMDP_NBLK=$nb ELEMDP_LDS_DEBUG=1 timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 > $O/eval_nblk$nb.txt 2>&1 || exit 1
  echo "nblk $nb: $(grep 'lin group' $O/eval_nblk$nb.txt | head -1 | sed 's/.*lds/lds/')"; grep "seq/s" $O/eval_nblk$nb.txt | tail -1
done
timeout -k 10 200 python tools/minibatch_bench.py 2000 200 40 > $O/mb.txt 2>&1; tail -2 $O/mb.txt
for c in 0 25 21 16 12; do
  e=""; [ $c -gt 0 ] && e="ELEMDP_CYK_CPB=$c"
  env $e timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan_cpb$c.txt 2>&1; echo "cyk cpb $c: $(tail -1 $O/scan_cpb$c.txt)"
done
