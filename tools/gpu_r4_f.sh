#!/bin/bash
# round 4: two blocks per workgroup at FULL residency (train LDS without the scan's position window; k4_out forced to 80 registers)
O=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_round4_gpu.py tests/test_round3_gpu.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
run() {  # name lib env...
  n=$1; L=$2; shift 2
  env ELEMDP_LIBRARY=$L ELEMDP_LDS_DEBUG=1 "$@" timeout -k 10 120 python tools/run_eval.py 4096 200 3 4 0 > $O/e_$n.txt 2>&1 || { echo "run failed: $n"; tail -3 $O/e_$n.txt; exit 1; }
  echo "$n: $(grep 'lin group' $O/e_$n.txt | head -1 | sed 's/.*with/with/') $(grep 'seq/s' $O/e_$n.txt | tail -1 | cut -c1-70)"
}
M=$GRAFT_REPO_ROOT/rnaelem_amd/libelemdp.so; R=$GRAFT_REPO_ROOT/build/var/lib_r1000.so
for rep in 1 2; do
run one_$rep $M ELEMDP_X=1
run mb22_$rep $M ELEMDP_NBLK=2
run mb22w6_$rep $M ELEMDP_NBLK=2 ELEMDP_MBW6=1
run mb22w6_r1000_$rep $R ELEMDP_NBLK=2 ELEMDP_MBW6=1
run mb12w6_$rep $M ELEMDP_NBLK_OUT=2 ELEMDP_MBW6=1
run mb33_$rep $M ELEMDP_NBLK=3
done
