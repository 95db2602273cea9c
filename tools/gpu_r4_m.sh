#!/bin/bash
# round 4: the round-3 library (git 9ccb784, built as build/var/lib_r3.so) against the current one on ONE box, alternating:
# the bench workload (10 000 x L=200), the scan (10 000 x L=300), the mini-batch iteration
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
cd $GRAFT_REPO_ROOT
R3=$GRAFT_REPO_ROOT/build/var/lib_r3.so; NOW=$GRAFT_REPO_ROOT/rnaelem_amd/libelemdp.so
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2 3; do
  for lib in r3 now; do
    L=$NOW; [ $lib = r3 ] && L=$R3
    ELEMDP_LIBRARY=$L timeout -k 10 200 $B > $O/b_${lib}_$rep.json 2> $O/b_${lib}_$rep.err || { echo "failed $lib"; tail -3 $O/b_${lib}_$rep.err; exit 1; }
    echo "bench $lib rep $rep: $(python -c "import json,sys; d=json.loads(open('$O/b_${lib}_$rep.json').read().strip().split(chr(10))[-1]); print('%.1f ms %.0f seq/s fn %.9g' % (d['ms_per_step'], d['value'], d['config']['fn']))")"
  done
done
for lib in r3 now r3 now; do
  L=$NOW; [ $lib = r3 ] && L=$R3
  ELEMDP_LIBRARY=$L timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan_$lib.txt 2>&1; echo "scan $lib: $(tail -1 $O/scan_$lib.txt)"
done
for lib in r3 now; do
  L=$NOW; [ $lib = r3 ] && L=$R3
  ELEMDP_LIBRARY=$L timeout -k 10 200 python tools/minibatch_bench.py 2000 200 40 > $O/mb_$lib.txt 2>&1; echo "mini-batch $lib: $(tail -1 $O/mb_$lib.txt)"
done
