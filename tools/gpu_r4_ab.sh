#!/bin/bash
# round 4: per-sequence filter kernels: resident sequences per CU (the kernels move ~7 TB/s of L2 misses: does the working set fit the Infinity Cache?)
O=$GRAFT_REPO_ROOT/gpurun_out/r4ab; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in main w4 t1024 t1024w8 t256; do
  L=$GRAFT_REPO_ROOT/build/var/lib_$lib.so; [ $lib = main ] && L=$GRAFT_REPO_ROOT/rnaelem_amd/libelemdp.so
  ELEMDP_LIBRARY=$L ELEMDP_TIME=1 timeout -k 10 200 python tools/load_laps.py 10000 300 > $O/laps_$lib.txt 2>&1 || { tail -5 $O/laps_$lib.txt; exit 1; }
  echo "$lib: $(grep 'BPP filter' $O/laps_$lib.txt | tail -3 | awk '{s+=$2} END {print s}') ms filter; $(grep '== load' $O/laps_$lib.txt | tail -1)"
done
done
