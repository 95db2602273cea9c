#!/bin/bash
# small-batch timing of prebuilt library variants (build/var/lib_*.so)
cd $GRAFT_REPO_ROOT
cp rnaelem_amd/libelemdp.so /tmp/keep.so
for f in build/var/lib_*.so; do
  cp $f rnaelem_amd/libelemdp.so
  echo "== $f"
  timeout -k 10 120 python tools/minibatch_bench.py 2000 200 30 --joint 2>&1 | tail -1
  timeout -k 10 120 python tools/run_eval.py 1250 200 3 4 2>&1 | tail -1
done
cp /tmp/keep.so rnaelem_amd/libelemdp.so
