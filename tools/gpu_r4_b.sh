#!/bin/bash
# round 4: record-area sizes (library variants of tools/build_variants.sh) x blocks per workgroup of k4_in / k4_out
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_round4_gpu.py -x -q -m gpu -k "lik or ranged" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
for lib in r768 r640 r512; do
  L=$GRAFT_REPO_ROOT/build/var/lib_$lib.so; [ $lib = main ] && L=$GRAFT_REPO_ROOT/rnaelem_amd/libelemdp.so
  for io in "3 3" "4 3" "3 2" "2 2" "4 4" "5 3"; do
    set -- $io
    ELEMDP_LIBRARY=$L ELEMDP_NBLK_IN=$1 ELEMDP_NBLK_OUT=$2 ELEMDP_LDS_DEBUG=1 timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 > $O/e_${lib}_$1_$2.txt 2>&1 || { echo "run failed: $lib $io"; tail -3 $O/e_${lib}_$1_$2.txt; exit 1; }
    echo "$lib in $1 out $2: $(grep 'lin group' $O/e_${lib}_$1_$2.txt | head -1 | sed 's/.*with/with/') $(grep 'seq/s' $O/e_${lib}_$1_$2.txt | tail -1 | cut -c1-70)"
  done
done
