#!/bin/bash
# round 4: what the floor of an evaluation is made of (workgroups that end behind their plan record: dbg 1031): kernel durations and
# launch gaps, two streams and one; then the 72-register (seven waves per SIMD) variants of k4_out
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ns in 2 1; do
  x=""; [ $ns = 1 ] && x="group_streams=1"
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt$ns -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 4096 200 2 4 0 dbg=1031 $x > $O/kt$ns.log 2>&1 || { tail -3 $O/kt$ns.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/kgaps.py $O/kt$ns > $O/gaps_streams$ns.txt; tail -2 $O/kt$ns.log | cut -c1-150; cat $O/gaps_streams$ns.txt
  rm -rf $O/kt$ns
done
cd $GRAFT_REPO_ROOT
M=$GRAFT_REPO_ROOT/rnaelem_amd/libelemdp.so
for rep in 1 2; do
  for lib in main v7a v7b; do
    L=$GRAFT_REPO_ROOT/build/var/lib_$lib.so; [ $lib = main ] && L=$M
    ELEMDP_LIBRARY=$L ELEMDP_LDS_DEBUG=1 timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 > $O/e_${lib}_$rep.txt 2>&1 || { echo "run failed: $lib"; tail -3 $O/e_${lib}_$rep.txt; exit 1; }
    echo "$lib: $(grep 'lin group' $O/e_${lib}_$rep.txt | head -1 | sed 's/.*lds/lds/' | cut -c1-40) $(grep 'seq/s' $O/e_${lib}_$rep.txt | tail -1 | cut -c1-70)"
  done
done
