#!/bin/bash
# round 4: compiler flag variants (SGPR spill traffic): timing at 4096 and the bench size
O=$GRAFT_REPO_ROOT/gpurun_out/r4p; mkdir -p $O
cd $GRAFT_REPO_ROOT
M=$GRAFT_REPO_ROOT/rnaelem_amd/libelemdp.so
for rep in 1 2 3; do
  for lib in main lsv0; do
    L=$GRAFT_REPO_ROOT/build/var/lib_$lib.so; [ $lib = main ] && L=$M
    ELEMDP_LIBRARY=$L timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 > $O/e_${lib}_$rep.txt 2>&1 || { echo "run failed: $lib"; tail -3 $O/e_${lib}_$rep.txt; exit 1; }
    echo "$lib: $(grep 'seq/s' $O/e_${lib}_$rep.txt | tail -1 | cut -c1-100)"
  done
done
