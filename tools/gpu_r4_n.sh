#!/bin/bash
# round 4: item sums with lane = (record, tuple) (variant t16) against one record per lane: parity of the variant, then timing
O=$GRAFT_REPO_ROOT/gpurun_out/r4n; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/build/var/lib_t16.so; K=$GRAFT_REPO_ROOT/build/var/lib_ktu1.so; M=$GRAFT_REPO_ROOT/rnaelem_amd/libelemdp.so
ELEMDP_LIBRARY=$T timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_round3_gpu.py tests/test_round4_gpu.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; grep -n "^E " $O/tests.log | head; exit $rc; fi
for rep in 1 2; do
  for lib in main t16 ktu1; do
    L=$M; [ $lib = t16 ] && L=$T; [ $lib = ktu1 ] && L=$K
    ELEMDP_LIBRARY=$L timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 > $O/e_${lib}_$rep.txt 2>&1 || { echo "run failed: $lib"; tail -3 $O/e_${lib}_$rep.txt; exit 1; }
    echo "$lib: $(grep 'seq/s' $O/e_${lib}_$rep.txt | tail -1 | cut -c1-100)"
  done
done
for lib in main t16; do
  L=$M; [ $lib = t16 ] && L=$T
  ELEMDP_LIBRARY=$L timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan_$lib.txt 2>&1; echo "scan $lib: $(tail -1 $O/scan_$lib.txt)"
done
