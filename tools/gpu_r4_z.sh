#!/bin/bash
# round 4: per-sequence filter kernels: what the inside sweep's phases cost (knock-out variants; kernel times from the trace)
O=$GRAFT_REPO_ROOT/gpurun_out/r4z; mkdir -p $O
for lib in k0 k1 k2 k4 k7; do
  cd /tmp && export TMPDIR=/tmp
  ELEMDP_LIBRARY=$GRAFT_REPO_ROOT/build/var/lib_$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$lib -o run -- python3 $GRAFT_REPO_ROOT/tools/load_laps.py 10000 300 > $O/kt_$lib.log 2>&1 || { tail -5 $O/kt_$lib.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  python tools/kstats.py $O/kt_$lib $O/ks_$lib.csv > /dev/null; rm -rf $O/kt_$lib
  echo "$lib: $(grep 'k6_in_seq\|k6_out_seq' $O/ks_$lib.csv | cut -d, -f1-4 | tr '\n' ' ')"
done
